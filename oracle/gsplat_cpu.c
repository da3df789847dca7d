/*
 * CPU ORACLE (test infrastructure, NOT product code) -- plain-C restatement of the gsplat v1.4.0 kernels
 * that MoBGS's render path executes, forward AND backward, scalar per-pixel loops, float32 like upstream.
 *
 * PARITY UNPINNED vs. real gsplat: gsplat==1.4.0 (/root/reference/README.md:26; imported at
 * /root/reference/gaussian_renderer/__init__.py:15) is an absent third-party dependency with no golden
 * vectors in the reference; this file restates the published algorithm of
 *   gsplat/cuda/csrc/fully_fused_projection_{fwd,bwd}.cu, isect_tiles.cu, rasterize_to_pixels_{fwd,bwd}.cu,
 *   utils.cuh (quat_to_rotmat, persp_proj(+_vjp), add_blur, inverse(+_vjp))           [SURVEY.md Appendix A]
 * and is itself cross-checked in tests/ against oracle/gsplat_torch.py (independent formulation, autograd
 * gradients).  Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC oracle/gsplat_cpu.c -o oracle/_build/libgsplat_cpu.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16

static void quat_to_rotmat(const float* q, float* R) {
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float inv = 1.0f / sqrtf(x * x + y * y + z * z + w * w);
    w *= inv; x *= inv; y *= inv; z *= inv;
    float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.f - 2.f * (y2 + z2); R[1] = 2.f * (xy - wz); R[2] = 2.f * (xz + wy);
    R[3] = 2.f * (xy + wz); R[4] = 1.f - 2.f * (x2 + z2); R[5] = 2.f * (yz - wx);
    R[6] = 2.f * (xz - wy); R[7] = 2.f * (yz + wx); R[8] = 1.f - 2.f * (x2 + y2);
}

static void mat3_mul(const float* A, const float* B, float* C) { /* C = A B */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}
static void mat3_mul_bt(const float* A, const float* B, float* C) { /* C = A B^T */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * j + k];
            C[3 * i + j] = s;
        }
}
static void mat3_mul_at(const float* A, const float* B, float* C) { /* C = A^T B */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += A[3 * k + i] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}

typedef struct {
    float R[9], t[3], fx, fy, cx, cy;
} cam_t;

static cam_t get_cam(const float* viewmats, const float* Ks, int c) {
    cam_t cam;
    const float* V = viewmats + 16 * c;
    const float* K = Ks + 9 * c;
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) cam.R[3 * r + k] = V[4 * r + k];
        cam.t[r] = V[4 * r + 3];
    }
    cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
    return cam;
}

typedef struct {
    float rz, rz2, tx, ty, lxp, lxn, lyp, lyn, J[6];
} persp_t;

static persp_t persp(const cam_t* cam, const float* p, int W, int H) {
    persp_t P;
    float tan_fovx = 0.5f * (float)W / cam->fx, tan_fovy = 0.5f * (float)H / cam->fy;
    P.lxp = ((float)W - cam->cx) / cam->fx + 0.3f * tan_fovx;
    P.lxn = cam->cx / cam->fx + 0.3f * tan_fovx;
    P.lyp = ((float)H - cam->cy) / cam->fy + 0.3f * tan_fovy;
    P.lyn = cam->cy / cam->fy + 0.3f * tan_fovy;
    P.rz = 1.f / p[2];
    P.rz2 = P.rz * P.rz;
    P.tx = p[2] * fminf(P.lxp, fmaxf(-P.lxn, p[0] * P.rz));
    P.ty = p[2] * fminf(P.lyp, fmaxf(-P.lyn, p[1] * P.rz));
    P.J[0] = cam->fx * P.rz; P.J[1] = 0.f; P.J[2] = -cam->fx * P.tx * P.rz2;
    P.J[3] = 0.f; P.J[4] = cam->fy * P.rz; P.J[5] = -cam->fy * P.ty * P.rz2;
    return P;
}

/* shared forward pieces: camera-space mean, covariance chain */
static void splat_chain(const cam_t* cam, const float* mean, const float* quat, const float* scale, float* p,
                        float* Rq, float* M, float* S3, float* Sc) {
    for (int r = 0; r < 3; ++r)
        p[r] = cam->R[3 * r] * mean[0] + cam->R[3 * r + 1] * mean[1] + cam->R[3 * r + 2] * mean[2] + cam->t[r];
    quat_to_rotmat(quat, Rq);
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) M[3 * r + k] = Rq[3 * r + k] * scale[k];
    mat3_mul_bt(M, M, S3);
    float RS[9];
    mat3_mul(cam->R, S3, RS);
    mat3_mul_bt(RS, cam->R, Sc);
}

void ora_project_fwd(int C, int N, const float* means, const float* quats, const float* scales,
                     const float* viewmats, const float* Ks, int W, int H, float eps2d, float near_plane,
                     float far_plane, float radius_clip, int32_t* radii, float* means2d, float* depths,
                     float* conics) {
    for (int c = 0; c < C; ++c) {
        cam_t cam = get_cam(viewmats, Ks, c);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i) {
            size_t o = (size_t)c * N + i;
            radii[o] = 0;
            means2d[2 * o] = means2d[2 * o + 1] = depths[o] = 0.f;
            conics[3 * o] = conics[3 * o + 1] = conics[3 * o + 2] = 0.f;
            float p[3], Rq[9], M[9], S3[9], Sc[9];
            splat_chain(&cam, means + 3 * i, quats + 4 * i, scales + 3 * i, p, Rq, M, S3, Sc);
            if (p[2] < near_plane || p[2] > far_plane) continue;
            persp_t P = persp(&cam, p, W, H);
            float JS[6];
            for (int r = 0; r < 2; ++r)
                for (int k = 0; k < 3; ++k)
                    JS[3 * r + k] = P.J[3 * r] * Sc[k] + P.J[3 * r + 1] * Sc[3 + k] + P.J[3 * r + 2] * Sc[6 + k];
            float a = JS[0] * P.J[0] + JS[1] * P.J[1] + JS[2] * P.J[2] + eps2d;
            float b = JS[0] * P.J[3] + JS[1] * P.J[4] + JS[2] * P.J[5];
            float d = JS[3] * P.J[3] + JS[4] * P.J[4] + JS[5] * P.J[5] + eps2d;
            float det = a * d - b * b;
            if (det <= 0.f) continue;
            float bb = 0.5f * (a + d);
            float v1 = bb + sqrtf(fmaxf(0.01f, bb * bb - det));
            float radius = ceilf(3.f * sqrtf(v1));
            if (radius <= radius_clip) continue;
            float x2d = cam.fx * p[0] * P.rz + cam.cx, y2d = cam.fy * p[1] * P.rz + cam.cy;
            if (x2d + radius <= 0 || x2d - radius >= W || y2d + radius <= 0 || y2d - radius >= H) continue;
            float inv = 1.f / det;
            radii[o] = (int32_t)radius;
            means2d[2 * o] = x2d; means2d[2 * o + 1] = y2d;
            depths[o] = p[2];
            conics[3 * o] = d * inv; conics[3 * o + 1] = -b * inv; conics[3 * o + 2] = a * inv;
        }
    }
}

/* v_viewmats accumulated in double; v_means/v_quats/v_scales summed over cameras */
void ora_project_bwd(int C, int N, const float* means, const float* quats, const float* scales,
                     const float* viewmats, const float* Ks, int W, int H, float eps2d, const int32_t* radii,
                     const float* conics, const float* v_means2d, const float* v_depths, const float* v_conics,
                     float* v_means, float* v_quats, float* v_scales, float* v_viewmats) {
    (void)eps2d;
    memset(v_means, 0, sizeof(float) * 3 * N);
    memset(v_quats, 0, sizeof(float) * 4 * N);
    memset(v_scales, 0, sizeof(float) * 3 * N);
    memset(v_viewmats, 0, sizeof(float) * 16 * C);
    for (int c = 0; c < C; ++c) {
        cam_t cam = get_cam(viewmats, Ks, c);
        double accR[9] = {0}, acct[3] = {0};
        for (int i = 0; i < N; ++i) {
            size_t o = (size_t)c * N + i;
            if (radii[o] <= 0) continue;
            const float* mean = means + 3 * i;
            const float* sc = scales + 3 * i;
            float p[3], Rq[9], M[9], S3[9], Sc[9];
            splat_chain(&cam, mean, quats + 4 * i, sc, p, Rq, M, S3, Sc);
            persp_t P = persp(&cam, p, W, H);
            /* inverse_vjp: v_S2 = -X G X, G = [[g0, g1/2],[g1/2, g2]] */
            float X[4] = {conics[3 * o], conics[3 * o + 1], conics[3 * o + 1], conics[3 * o + 2]};
            float G[4] = {0, 0, 0, 0};
            if (v_conics) { G[0] = v_conics[3 * o]; G[1] = G[2] = 0.5f * v_conics[3 * o + 1]; G[3] = v_conics[3 * o + 2]; }
            float XG[4] = {X[0] * G[0] + X[1] * G[2], X[0] * G[1] + X[1] * G[3], X[2] * G[0] + X[3] * G[2], X[2] * G[1] + X[3] * G[3]};
            float vS[4] = {-(XG[0] * X[0] + XG[1] * X[2]), -(XG[0] * X[1] + XG[1] * X[3]),
                           -(XG[2] * X[0] + XG[3] * X[2]), -(XG[2] * X[1] + XG[3] * X[3])};
            /* persp_proj_vjp */
            float vSc[9], vJ[6];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    float s = 0.f;
                    for (int r = 0; r < 2; ++r)
                        for (int q = 0; q < 2; ++q) s += P.J[3 * r + a] * vS[2 * r + q] * P.J[3 * q + b];
                    vSc[3 * a + b] = s;
                }
            for (int r = 0; r < 2; ++r)
                for (int k = 0; k < 3; ++k) {
                    float s = 0.f;
                    for (int q = 0; q < 2; ++q)
                        for (int l = 0; l < 3; ++l)
                            s += vS[2 * r + q] * P.J[3 * q + l] * Sc[3 * k + l] + vS[2 * q + r] * P.J[3 * q + l] * Sc[3 * l + k];
                    vJ[3 * r + k] = s;
                }
            float gx = v_means2d ? v_means2d[2 * o] : 0.f, gy = v_means2d ? v_means2d[2 * o + 1] : 0.f;
            float rz = P.rz, rz2 = P.rz2, rz3 = rz2 * rz;
            float vp[3];
            vp[0] = cam.fx * rz * gx;
            vp[1] = cam.fy * rz * gy;
            vp[2] = -(cam.fx * p[0] * gx + cam.fy * p[1] * gy) * rz2;
            if (p[0] * rz <= P.lxp && p[0] * rz >= -P.lxn) vp[0] += -cam.fx * rz2 * vJ[2];
            else vp[2] += -cam.fx * rz3 * vJ[2] * P.tx;
            if (p[1] * rz <= P.lyp && p[1] * rz >= -P.lyn) vp[1] += -cam.fy * rz2 * vJ[5];
            else vp[2] += -cam.fy * rz3 * vJ[5] * P.ty;
            vp[2] += -cam.fx * rz2 * vJ[0] - cam.fy * rz2 * vJ[4] + 2.f * cam.fx * P.tx * rz3 * vJ[2] + 2.f * cam.fy * P.ty * rz3 * vJ[5];
            if (v_depths) vp[2] += v_depths[o];
            /* pos_world_to_cam_vjp + covar_world_to_cam_vjp */
            float vR[9];
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) vR[3 * r + k] = vp[r] * mean[k];
            for (int k = 0; k < 3; ++k) v_means[3 * i + k] += cam.R[k] * vp[0] + cam.R[3 + k] * vp[1] + cam.R[6 + k] * vp[2];
            float RS3t[9], RS3[9], T1[9], T2[9];
            mat3_mul_bt(cam.R, S3, RS3t);
            mat3_mul(cam.R, S3, RS3);
            mat3_mul(vSc, RS3t, T1);
            mat3_mul_at(vSc, RS3, T2);
            for (int k = 0; k < 9; ++k) vR[k] += T1[k] + T2[k];
            float RtV[9], vS3[9];
            mat3_mul_at(cam.R, vSc, RtV);
            mat3_mul(RtV, cam.R, vS3);
            /* quat_scale_to_covar_vjp */
            float Sy[9], vM[9];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) Sy[3 * a + b] = vS3[3 * a + b] + vS3[3 * b + a];
            mat3_mul(Sy, M, vM);
            float vRq[9];
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) vRq[3 * r + k] = vM[3 * r + k] * sc[k];
            for (int k = 0; k < 3; ++k) v_scales[3 * i + k] += Rq[k] * vM[k] + Rq[3 + k] * vM[3 + k] + Rq[6 + k] * vM[6 + k];
            const float* q = quats + 4 * i;
            float inv = 1.f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
            float vn[4];
            vn[0] = 2.f * (x * (vRq[7] - vRq[5]) + y * (vRq[2] - vRq[6]) + z * (vRq[3] - vRq[1]));
            vn[1] = 2.f * (-2.f * x * (vRq[4] + vRq[8]) + y * (vRq[1] + vRq[3]) + z * (vRq[2] + vRq[6]) + w * (vRq[7] - vRq[5]));
            vn[2] = 2.f * (x * (vRq[1] + vRq[3]) - 2.f * y * (vRq[0] + vRq[8]) + z * (vRq[5] + vRq[7]) + w * (vRq[2] - vRq[6]));
            vn[3] = 2.f * (x * (vRq[2] + vRq[6]) + y * (vRq[5] + vRq[7]) - 2.f * z * (vRq[0] + vRq[4]) + w * (vRq[3] - vRq[1]));
            float dotp = vn[0] * w + vn[1] * x + vn[2] * y + vn[3] * z;
            v_quats[4 * i] += (vn[0] - dotp * w) * inv;
            v_quats[4 * i + 1] += (vn[1] - dotp * x) * inv;
            v_quats[4 * i + 2] += (vn[2] - dotp * y) * inv;
            v_quats[4 * i + 3] += (vn[3] - dotp * z) * inv;
            for (int k = 0; k < 9; ++k) accR[k] += vR[k];
            for (int k = 0; k < 3; ++k) acct[k] += vp[k];
        }
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k) v_viewmats[16 * c + 4 * r + k] = (float)accR[3 * r + k];
            v_viewmats[16 * c + 4 * r + 3] = (float)acct[r];
        }
    }
}

/* ---- isect_tiles + sort + offsets ---------------------------------------------------------------- */
static void tile_rect(float mx, float my, int radius, int tw, int th, int* x0, int* y0, int* x1, int* y1) {
    float tr = (float)radius / TILE, tx = mx / TILE, ty = my / TILE;
    float fx0 = floorf(tx - tr), fx1 = ceilf(tx + tr), fy0 = floorf(ty - tr), fy1 = ceilf(ty + tr);
    *x0 = (int)fminf(fmaxf(fx0, 0.f), (float)tw);
    *x1 = (int)fminf(fmaxf(fx1, 0.f), (float)tw);
    *y0 = (int)fminf(fmaxf(fy0, 0.f), (float)th);
    *y1 = (int)fminf(fmaxf(fy1, 0.f), (float)th);
}

typedef struct {
    uint64_t key;
    int32_t id;
} isect_t;

static int cmp_isect(const void* a, const void* b) {
    const isect_t* x = (const isect_t*)a;
    const isect_t* y = (const isect_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return (x->id > y->id) - (x->id < y->id); /* stable radix sort == ties by emission order == ascending id */
}

/* pass 1: tiles_per_gauss, returns total */
int64_t ora_isect_count(int C, int N, int W, int H, const float* means2d, const int32_t* radii,
                        int32_t* tiles_per_gauss) {
    int tw = (W + TILE - 1) / TILE, th = (H + TILE - 1) / TILE;
    int64_t total = 0;
    for (size_t o = 0; o < (size_t)C * N; ++o) {
        int n = 0;
        if (radii[o] > 0) {
            int x0, y0, x1, y1;
            tile_rect(means2d[2 * o], means2d[2 * o + 1], radii[o], tw, th, &x0, &y0, &x1, &y1);
            n = (x1 - x0) * (y1 - y0);
        }
        tiles_per_gauss[o] = n;
        total += n;
    }
    return total;
}

/* pass 2: sorted isect_ids / flatten_ids + isect_offsets [C*th*tw] */
void ora_isect_sort(int C, int N, int W, int H, const float* means2d, const int32_t* radii, const float* depths,
                    int64_t n_isects, int64_t* isect_ids, int32_t* flatten_ids, int32_t* isect_offsets) {
    int tw = (W + TILE - 1) / TILE, th = (H + TILE - 1) / TILE;
    int n_tiles = tw * th;
    int tile_bits = 0;
    while ((1ll << tile_bits) <= (long long)n_tiles) ++tile_bits; /* floor(log2(n_tiles)) + 1 */
    isect_t* buf = (isect_t*)malloc(sizeof(isect_t) * (size_t)(n_isects > 0 ? n_isects : 1));
    int64_t k = 0;
    for (size_t o = 0; o < (size_t)C * N; ++o) {
        if (radii[o] <= 0) continue;
        int x0, y0, x1, y1;
        tile_rect(means2d[2 * o], means2d[2 * o + 1], radii[o], tw, th, &x0, &y0, &x1, &y1);
        uint32_t db;
        memcpy(&db, &depths[o], 4);
        uint64_t cam = o / N;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                uint64_t tile = (uint64_t)y * tw + x;
                buf[k].key = (cam << (32 + tile_bits)) | (tile << 32) | db;
                buf[k].id = (int32_t)o;
                ++k;
            }
    }
    qsort(buf, (size_t)k, sizeof(isect_t), cmp_isect);
    for (int64_t j = 0; j < k; ++j) {
        isect_ids[j] = (int64_t)buf[j].key;
        flatten_ids[j] = buf[j].id;
    }
    /* isect_offset_encode */
    int64_t j = 0;
    for (int t = 0; t < C * n_tiles; ++t) {
        uint64_t cam = t / n_tiles, tile = t % n_tiles;
        uint64_t lo = ((cam << tile_bits) | tile);
        while (j < k && (buf[j].key >> 32) < lo) ++j;
        isect_offsets[t] = (int32_t)j;
    }
    free(buf);
}

/* ---- rasterize_to_pixels forward ------------------------------------------------------------------- */
void ora_raster_fwd(int C, int N, int D, int W, int H, const float* means2d, const float* conics,
                    const float* colors /* [C*N, D] */, const float* opacities /* [C*N] */,
                    const float* backgrounds /* [C,D] or NULL */, const int32_t* isect_offsets,
                    const int32_t* flatten_ids, int64_t n_isects, float* render, float* alphas, int32_t* last_ids) {
    (void)N;
    int tw = (W + TILE - 1) / TILE, th = (H + TILE - 1) / TILE;
    int n_tiles = tw * th;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < C * n_tiles; ++t) {
        int cam = t / n_tiles, tl = t % n_tiles, ty = tl / tw, tx = tl % tw;
        int s = isect_offsets[t];
        int e = (t == C * n_tiles - 1) ? (int)n_isects : isect_offsets[t + 1];
        for (int i = ty * TILE; i < (ty + 1) * TILE && i < H; ++i)
            for (int j = tx * TILE; j < (tx + 1) * TILE && j < W; ++j) {
                float px = (float)j + 0.5f, py = (float)i + 0.5f;
                float T = 1.f;
                float pix[64];
                for (int k = 0; k < D; ++k) pix[k] = 0.f;
                int cur = 0;
                for (int idx = s; idx < e; ++idx) {
                    int g = flatten_ids[idx];
                    float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
                    float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
                    float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                    float alpha = fminf(0.999f, opacities[g] * expf(-sigma));
                    if (sigma < 0.f || alpha < 1.f / 255.f) continue;
                    float nT = T * (1.f - alpha);
                    if (nT <= 1e-4f) break;
                    float vis = alpha * T;
                    const float* c = colors + (size_t)g * D;
                    for (int k = 0; k < D; ++k) pix[k] += c[k] * vis;
                    cur = idx;
                    T = nT;
                }
                size_t p = ((size_t)cam * H + i) * W + j;
                alphas[p] = 1.f - T;
                last_ids[p] = cur;
                for (int k = 0; k < D; ++k)
                    render[p * D + k] = backgrounds ? pix[k] + T * backgrounds[cam * D + k] : pix[k];
            }
    }
}

/* ---- rasterize_to_pixels backward -------------------------------------------------------------------
 * Per tile the per-splat gradients are accumulated in a tile-local float buffer (upstream: warp reduction),
 * then added to the global arrays with atomic float adds (upstream: atomicAdd) -- order differs run to run. */
static inline void atomic_addf(float* p, float v) {
#pragma omp atomic
    *p += v;
}

void ora_raster_bwd(int C, int N, int D, int W, int H, const float* means2d, const float* conics,
                    const float* colors, const float* opacities, const float* backgrounds,
                    const int32_t* isect_offsets, const int32_t* flatten_ids, int64_t n_isects,
                    const float* render_alphas, const int32_t* last_ids, const float* v_render,
                    const float* v_alphas, float* v_means2d, float* v_conics, float* v_colors, float* v_opacities) {
    int tw = (W + TILE - 1) / TILE, th = (H + TILE - 1) / TILE;
    int n_tiles = tw * th;
    size_t G = (size_t)C * N;
    int stride = 6 + D;
    memset(v_means2d, 0, sizeof(float) * 2 * G);
    memset(v_conics, 0, sizeof(float) * 3 * G);
    memset(v_colors, 0, sizeof(float) * D * G);
    memset(v_opacities, 0, sizeof(float) * G);
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < C * n_tiles; ++t) {
        int cam = t / n_tiles, tl = t % n_tiles, ty = tl / tw, tx = tl % tw;
        int s = isect_offsets[t];
        int e = (t == C * n_tiles - 1) ? (int)n_isects : isect_offsets[t + 1];
        if (e <= s) continue;
        float* loc = (float*)calloc((size_t)(e - s) * stride, sizeof(float));
        for (int i = ty * TILE; i < (ty + 1) * TILE && i < H; ++i)
            for (int j = tx * TILE; j < (tx + 1) * TILE && j < W; ++j) {
                size_t p = ((size_t)cam * H + i) * W + j;
                float px = (float)j + 0.5f, py = (float)i + 0.5f;
                float T_final = 1.f - render_alphas[p];
                float T = T_final;
                float buffer[64];
                for (int k = 0; k < D; ++k) buffer[k] = 0.f;
                int bin_final = last_ids[p];
                const float* vr = v_render + p * D;
                float va = v_alphas ? v_alphas[p] : 0.f;
                int top = bin_final < e - 1 ? bin_final : e - 1;
                for (int idx = top; idx >= s; --idx) {
                    int g = flatten_ids[idx];
                    float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
                    float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
                    float opac = opacities[g];
                    float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                    float vis = expf(-sigma);
                    float alpha = fminf(0.999f, opac * vis);
                    if (sigma < 0.f || alpha < 1.f / 255.f) continue;
                    float ra = 1.f / (1.f - alpha);
                    T *= ra;
                    float fac = alpha * T;
                    float* a = loc + (size_t)(idx - s) * stride;
                    const float* c = colors + (size_t)g * D;
                    float v_alpha = 0.f;
                    for (int k = 0; k < D; ++k) {
                        a[6 + k] += fac * vr[k];
                        v_alpha += (c[k] * T - buffer[k] * ra) * vr[k];
                    }
                    v_alpha += T_final * ra * va;
                    if (backgrounds) {
                        float accum = 0.f;
                        for (int k = 0; k < D; ++k) accum += backgrounds[cam * D + k] * vr[k];
                        v_alpha += -T_final * ra * accum;
                    }
                    if (opac * vis <= 0.999f) {
                        float v_sigma = -opac * vis * v_alpha;
                        a[0] += v_sigma * (ca * dx + cb * dy);
                        a[1] += v_sigma * (cb * dx + cc * dy);
                        a[2] += 0.5f * v_sigma * dx * dx;
                        a[3] += v_sigma * dx * dy;
                        a[4] += 0.5f * v_sigma * dy * dy;
                        a[5] += vis * v_alpha;
                    }
                    for (int k = 0; k < D; ++k) buffer[k] += c[k] * fac;
                }
            }
        for (int idx = s; idx < e; ++idx) {
            const float* a = loc + (size_t)(idx - s) * stride;
            size_t g = (size_t)flatten_ids[idx];
            atomic_addf(&v_means2d[2 * g], a[0]);
            atomic_addf(&v_means2d[2 * g + 1], a[1]);
            atomic_addf(&v_conics[3 * g], a[2]);
            atomic_addf(&v_conics[3 * g + 1], a[3]);
            atomic_addf(&v_conics[3 * g + 2], a[4]);
            atomic_addf(&v_opacities[g], a[5]);
            for (int k = 0; k < D; ++k) atomic_addf(&v_colors[g * D + k], a[6 + k]);
        }
        free(loc);
    }
}

int ora_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void ora_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
