// K1 / K2: per-splat perspective projection (EWA) forward and backward for gfx950.
//
// Semantics follow gsplat v1.4.0 fully_fused_projection_{fwd,bwd}.cu + utils.cuh [upstream, restated in
// SURVEY.md Appendix A.1]; called by the reference at /root/reference/gaussian_renderer/__init__.py:190-199
// and inside every rasterization() call (:143, :163, :201, ...).
//
// Streaming kernels, one thread per (camera, splat): 44 B in + 32 B out per splat forward,
// 100 B in + 40 B out backward.  HBM-bound; the camera (R, t, K) is wave-uniform and lives in SGPRs.
#include <stdarg.h>

#include "common.h"
#include "prep_shared.h"

namespace mobgs {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct Cam {
    float R[9];
    float t[3];
    float fx, fy, cx, cy;
};

__device__ inline Cam load_cam(const float* __restrict__ viewmats, const float* __restrict__ Ks, int c) {
    Cam cam;
    const float* V = viewmats + 16 * c;
    const float* K = Ks + 9 * c;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        cam.R[3 * r + 0] = V[4 * r + 0];
        cam.R[3 * r + 1] = V[4 * r + 1];
        cam.R[3 * r + 2] = V[4 * r + 2];
        cam.t[r] = V[4 * r + 3];
    }
    cam.fx = K[0];
    cam.fy = K[4];
    cam.cx = K[2];
    cam.cy = K[5];
    return cam;
}

// rotation matrix of the NORMALISED quaternion (w,x,y,z); row-major
__device__ inline void quat_to_rotmat(const float q[4], float R[9]) {
    // 1 / sqrt in the oracles' order and with correctly rounded operations (v_rsq_f32 is ~1 ulp: the last bit of the
    // rotation decided the radius = ceil(3 sqrt(lambda_max)) of one splat in ~8000 of an extreme scene differently from
    // the oracles, scripts/soak_parity.py; upstream's own CUDA rsqrtf cannot be reproduced bit for bit either way)
    const float inv = 1.f / sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + q[0] * q[0]);
    const float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    const float xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.f - 2.f * (y2 + z2);
    R[1] = 2.f * (xy - wz);
    R[2] = 2.f * (xz + wy);
    R[3] = 2.f * (xy + wz);
    R[4] = 1.f - 2.f * (x2 + z2);
    R[5] = 2.f * (yz - wx);
    R[6] = 2.f * (xz - wy);
    R[7] = 2.f * (yz + wx);
    R[8] = 1.f - 2.f * (x2 + y2);
}

// C = A * B (3x3 row-major)
__device__ inline void mm3(const float* A, const float* B, float* Cm) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Cm[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// C = A * B^T
__device__ inline void mm3_abt(const float* A, const float* B, float* Cm) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Cm[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
// C = A^T * B
__device__ inline void mm3_atb(const float* A, const float* B, float* Cm) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Cm[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

struct Persp {
    float rz, rz2, tx, ty;
    float lim_x_pos, lim_x_neg, lim_y_pos, lim_y_neg;
    float J[6];  // 2x3 row-major
};

__device__ inline Persp persp_setup(const Cam& cam, const float p[3], int width, int height) {
    Persp P;
    const float tan_fovx = 0.5f * (float)width / cam.fx;
    const float tan_fovy = 0.5f * (float)height / cam.fy;
    P.lim_x_pos = ((float)width - cam.cx) / cam.fx + 0.3f * tan_fovx;
    P.lim_x_neg = cam.cx / cam.fx + 0.3f * tan_fovx;
    P.lim_y_pos = ((float)height - cam.cy) / cam.fy + 0.3f * tan_fovy;
    P.lim_y_neg = cam.cy / cam.fy + 0.3f * tan_fovy;
    P.rz = 1.f / p[2];
    P.rz2 = P.rz * P.rz;
    P.tx = p[2] * fminf(P.lim_x_pos, fmaxf(-P.lim_x_neg, p[0] * P.rz));
    P.ty = p[2] * fminf(P.lim_y_pos, fmaxf(-P.lim_y_neg, p[1] * P.rz));
    P.J[0] = cam.fx * P.rz;
    P.J[1] = 0.f;
    P.J[2] = -cam.fx * P.tx * P.rz2;
    P.J[3] = 0.f;
    P.J[4] = cam.fy * P.rz;
    P.J[5] = -cam.fy * P.ty * P.rz2;
    return P;
}

// PREP (mobgs_prep_project_and_bin_fused, one camera): the thread BUILDS its splat's state from the raw parameters
// (prep_shared.h: spline position, rotation, exp / sigmoid activations, colour features -- the arithmetic of
// prep_fwd_kernel, bit for bit) instead of loading it, and writes position / rotation / scales / opacity out for the
// backward pass and the caller's result dict; the colour features go straight into the compositor's record.
struct PrepFused {
    PrepIn<float> in;
    float *means, *quats, *scales, *opac;
};
#ifndef MOBGS_PROJ_FWD_THREADS
#define MOBGS_PROJ_FWD_THREADS 256
#endif
template <bool PREP>
__global__ void __launch_bounds__(MOBGS_PROJ_FWD_THREADS)
project_fwd_kernel(int N, const float* __restrict__ means, const float* __restrict__ quats,
                   const float* __restrict__ scales, const float* __restrict__ viewmats,
                   const float* __restrict__ Ks, int width, int height, float eps2d, float near_plane,
                   float far_plane, float radius_clip, int tile_w, int tile_h,
                   int32_t* __restrict__ radii, float* __restrict__ means2d, float* __restrict__ depths,
                   float* __restrict__ conics, int32_t* __restrict__ tiles_per_gauss, int32_t* __restrict__ zero_ptr,
                   unsigned zero_n, PackArgs pack, int geom_stride, BinArgs bin, PrepFused prep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    // geometry_per_camera: camera c reads rows [c N, c N + N) of means / quats (geom_stride = N), else the shared rows
    means += (size_t)3 * c * geom_stride;
    quats += (size_t)4 * c * geom_stride;
    // side job for the orchestrator: clear the binning counters (saves a memset launch on the critical path)
    if (zero_ptr && c == 0)
        for (unsigned z = (unsigned)i; z < zero_n; z += gridDim.x * blockDim.x) zero_ptr[z] = 0;
    if (i >= N) return;
    const Cam cam = load_cam(viewmats, Ks, c);
    const size_t o = (size_t)c * N + i;

    float m[3], pq[4] = {0.f, 0.f, 0.f, 0.f}, ps[3] = {0.f, 0.f, 0.f}, pop = 0.f, pcol[9];
    if constexpr (PREP) {
        prep_splat(prep.in, i, m, pq, ps, pop, pcol);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            prep.means[3 * i + k] = m[k];
            prep.scales[3 * i + k] = ps[k];
        }
        reinterpret_cast<float4*>(prep.quats)[i] = make_float4(pq[0], pq[1], pq[2], pq[3]);
        prep.opac[i] = pop;
    } else {
        m[0] = means[3 * i];
        m[1] = means[3 * i + 1];
        m[2] = means[3 * i + 2];
    }
    float p[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        p[r] = cam.R[3 * r] * m[0] + cam.R[3 * r + 1] * m[1] + cam.R[3 * r + 2] * m[2] + cam.t[r];

    int rad = 0;
    float m2x = 0.f, m2y = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, depth = 0.f;
    int ntiles = 0;
    TileRect tr{0, 0, 0, 0};
    if (p[2] >= near_plane && p[2] <= far_plane) {
        float q[4], s[3];
        if constexpr (PREP) {
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = pq[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) s[k] = ps[k];
        } else {
            const float4 qv = reinterpret_cast<const float4*>(quats)[i];
            q[0] = qv.x; q[1] = qv.y; q[2] = qv.z; q[3] = qv.w;
            s[0] = scales[3 * i]; s[1] = scales[3 * i + 1]; s[2] = scales[3 * i + 2];
        }
        float Rq[9], M[9], S3[9], RS[9], Sc[9];
        quat_to_rotmat(q, Rq);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            M[3 * r] = Rq[3 * r] * s[0];
            M[3 * r + 1] = Rq[3 * r + 1] * s[1];
            M[3 * r + 2] = Rq[3 * r + 2] * s[2];
        }
        mm3_abt(M, M, S3);       // Sigma3 = M M^T
        mm3(cam.R, S3, RS);      // R Sigma3
        mm3_abt(RS, cam.R, Sc);  // R Sigma3 R^T
        const Persp P = persp_setup(cam, p, width, height);
        // Sigma2 = J Sc J^T
        float JS[6];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                JS[3 * r + k] = P.J[3 * r] * Sc[k] + P.J[3 * r + 1] * Sc[3 + k] + P.J[3 * r + 2] * Sc[6 + k];
        const float a = JS[0] * P.J[0] + JS[1] * P.J[1] + JS[2] * P.J[2] + eps2d;
        const float b = JS[0] * P.J[3] + JS[1] * P.J[4] + JS[2] * P.J[5];
        const float d = JS[3] * P.J[3] + JS[4] * P.J[4] + JS[5] * P.J[5] + eps2d;
        const float det = a * d - b * b;
        if (det > 0.f) {
            const float bb = 0.5f * (a + d);
            const float v1 = bb + sqrtf(fmaxf(0.01f, bb * bb - det));
            const float radius = ceilf(3.f * sqrtf(v1));
            const float x2d = cam.fx * p[0] * P.rz + cam.cx;
            const float y2d = cam.fy * p[1] * P.rz + cam.cy;
            const bool outside = (x2d + radius <= 0.f) || (x2d - radius >= (float)width) ||
                                 (y2d + radius <= 0.f) || (y2d - radius >= (float)height);
            if (radius > radius_clip && !outside) {
                const float inv_det = 1.f / det;
                rad = (int)radius;
                m2x = x2d;
                m2y = y2d;
                ca = d * inv_det;
                cb = -b * inv_det;
                cc = a * inv_det;
                depth = p[2];
                tr = tile_rect(m2x, m2y, rad, tile_w, tile_h);
                ntiles = (tr.x1 - tr.x0) * (tr.y1 - tr.y0);
            }
        }
    }
    radii[o] = rad;
    reinterpret_cast<float2*>(means2d)[o] = make_float2(m2x, m2y);
    depths[o] = depth;
    conics[3 * o] = ca;
    conics[3 * o + 1] = cb;
    conics[3 * o + 2] = cc;
    tiles_per_gauss[o] = ntiles;
    // side job for the orchestrator: the compositor's record of this splat (depth as the extra channel) while
    // everything it needs is in registers -- saves the pack launch and re-reading means2d / conics / depths
    // ... and its bin record (fused binning path, common.h): only splats with tiles are ever looked up
    if constexpr (PREP) {
        if (bin.records && ntiles > 0)
            write_bin_record(bin.records + o * BIN_RECORD_FLOATS, m2x, m2y, ca, cb, cc, depth, pop, bin.cull, tr, c);
        if (pack.records && rad > 0)
            write_splat_record(pack.records + o * pack.stride, m2x, m2y, ca, cb, cc, pop, pcol, 9, true, depth);
    } else {
        if (bin.records && ntiles > 0)
            write_bin_record(bin.records + o * BIN_RECORD_FLOATS, m2x, m2y, ca, cb, cc, depth,
                             bin.opacities[bin.opac_per_camera ? o : (size_t)i], bin.cull, tr, c);
        if (pack.records && rad > 0)
            write_splat_record(pack.records + o * pack.stride, m2x, m2y, ca, cb, cc,
                               pack.opacities[pack.opac_per_camera ? o : (size_t)i],
                               pack.colors + (pack.colors_per_camera ? o : (size_t)i) * pack.channels, pack.channels, true,
                               depth);
    }
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// PREPB (mobgs_project_prep_bwd_fused, one camera): the thread hands its splat's state cotangents -- in registers -- to
// the prep backward (prep_shared.h prep_bwd_apply: the arithmetic of prep_bwd_kernel, bit for bit) instead of storing
// v_means / v_quats / v_scales for a second launch to read back; the leaf gradients leave this kernel.
struct LeafGrads {
    float *s_xyz, *s_scaling, *s_rotation, *s_opacity, *s_fdc, *s_ft, *d_control, *d_scaling, *d_rotation, *d_omega,
        *d_opacity, *d_fdc, *d_ft;
};
struct PrepBwdFused {
    int Ns, Nd, accumulate;
    const float* times;
    const long long* d_ncp;
    const float *d_trbf, *opac, *v_opac, *v_colors;
    const float *x_means, *x_quats, *x_scales;  // cotangents that reach the state directly (optional)
    LeafGrads g;
};
template <bool PREPB>
__global__ void __launch_bounds__(256)
project_bwd_kernel(int N, const float* __restrict__ means, const float* __restrict__ quats,
                   const float* __restrict__ scales, const float* __restrict__ viewmats,
                   const float* __restrict__ Ks, int width, int height, float eps2d,
                   const int32_t* __restrict__ radii, const float* __restrict__ conics,
                   const float* __restrict__ v_means2d, const float* __restrict__ v_depths,
                   const float* __restrict__ v_conics, float* __restrict__ v_means,
                   float* __restrict__ v_quats, float* __restrict__ v_scales,
                   float* __restrict__ v_view_partial, int accumulate, int accumulate_scales, size_t geom_stride,
                   size_t scales_stride, PrepBwdFused pb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if constexpr (PREPB) {
        if (!pb.accumulate) prep_bwd_clear_rows(pb.Ns, pb.Nd, pb.g.d_control);
    }
    // cameras in parallel (grid.y = C > 1, per-camera geometry): camera c reads / writes its own rows of means, quats
    // and their gradients, and its own copy of the shared scales' gradient (summed over the cameras afterwards)
    means += 3 * c * geom_stride;
    quats += 4 * c * geom_stride;
    v_means += 3 * c * geom_stride;
    v_quats += 4 * c * geom_stride;
    v_scales += 3 * c * scales_stride;
    const Cam cam = load_cam(viewmats, Ks, c);
    float vR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float vt[3] = {0.f, 0.f, 0.f};
    float vm[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
    const size_t o = (size_t)c * N + (i < N ? i : 0);
    const bool live = (i < N) && radii[o] > 0;
    if (live) {
        const float m[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
        float p[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            p[r] = cam.R[3 * r] * m[0] + cam.R[3 * r + 1] * m[1] + cam.R[3 * r + 2] * m[2] + cam.t[r];
        const float4 qv = reinterpret_cast<const float4*>(quats)[i];
        const float qraw[4] = {qv.x, qv.y, qv.z, qv.w};
        const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        float Rq[9], M[9], S3[9], RS[9], Sc[9];
        quat_to_rotmat(qraw, Rq);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            M[3 * r] = Rq[3 * r] * s[0];
            M[3 * r + 1] = Rq[3 * r + 1] * s[1];
            M[3 * r + 2] = Rq[3 * r + 2] * s[2];
        }
        mm3_abt(M, M, S3);
        mm3(cam.R, S3, RS);
        mm3_abt(RS, cam.R, Sc);
        const Persp P = persp_setup(cam, p, width, height);

        // --- conic = inverse(Sigma2'):  v_Sigma2 = -X v_X X with X = conic matrix (inverse_vjp)
        const float X0 = conics[3 * o], X1 = conics[3 * o + 1], X2 = conics[3 * o + 2];
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (v_conics) {
            g0 = v_conics[3 * o];
            g1 = 0.5f * v_conics[3 * o + 1];
            g2 = v_conics[3 * o + 2];
        }
        // A = X * G
        const float A00 = X0 * g0 + X1 * g1, A01 = X0 * g1 + X1 * g2;
        const float A10 = X1 * g0 + X2 * g1, A11 = X1 * g1 + X2 * g2;
        // vS2 = -(A * X)   (2x2, symmetric up to rounding)
        const float vS00 = -(A00 * X0 + A01 * X1), vS01 = -(A00 * X1 + A01 * X2);
        const float vS10 = -(A10 * X0 + A11 * X1), vS11 = -(A10 * X1 + A11 * X2);

        // --- Sigma2 = J Sc J^T
        // v_Sc = J^T vS2 J (3x3);  v_J = vS2 J Sc^T + vS2^T J Sc
        float vSJ[6];  // vS2 * J  (2x3)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            vSJ[k] = vS00 * P.J[k] + vS01 * P.J[3 + k];
            vSJ[3 + k] = vS10 * P.J[k] + vS11 * P.J[3 + k];
        }
        float vSc[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) vSc[3 * a + b] = P.J[a] * vSJ[b] + P.J[3 + a] * vSJ[3 + b];
        float vStJ[6];  // vS2^T * J
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            vStJ[k] = vS00 * P.J[k] + vS10 * P.J[3 + k];
            vStJ[3 + k] = vS01 * P.J[k] + vS11 * P.J[3 + k];
        }
        float vJ[6];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // (vS2 J) Sc^T [r][k] = sum_l vSJ[r][l] Sc[k][l];  (vS2^T J) Sc [r][k] = sum_l vStJ[r][l] Sc[l][k]
                vJ[3 * r + k] = vSJ[3 * r] * Sc[3 * k] + vSJ[3 * r + 1] * Sc[3 * k + 1] + vSJ[3 * r + 2] * Sc[3 * k + 2] +
                                vStJ[3 * r] * Sc[k] + vStJ[3 * r + 1] * Sc[3 + k] + vStJ[3 * r + 2] * Sc[6 + k];
            }

        // --- mean2d, depth and J as functions of p (persp_proj_vjp)
        float vp[3] = {0.f, 0.f, 0.f};
        float gx = 0.f, gy = 0.f;
        if (v_means2d) {
            gx = v_means2d[2 * o];
            gy = v_means2d[2 * o + 1];
        }
        const float rz = P.rz, rz2 = P.rz2, rz3 = rz2 * rz;
        vp[0] += cam.fx * rz * gx;
        vp[1] += cam.fy * rz * gy;
        vp[2] += -(cam.fx * p[0] * gx + cam.fy * p[1] * gy) * rz2;
        const float xr = p[0] * rz, yr = p[1] * rz;
        if (xr <= P.lim_x_pos && xr >= -P.lim_x_neg)
            vp[0] += -cam.fx * rz2 * vJ[2];
        else
            vp[2] += -cam.fx * rz3 * vJ[2] * P.tx;
        if (yr <= P.lim_y_pos && yr >= -P.lim_y_neg)
            vp[1] += -cam.fy * rz2 * vJ[5];
        else
            vp[2] += -cam.fy * rz3 * vJ[5] * P.ty;
        vp[2] += -cam.fx * rz2 * vJ[0] - cam.fy * rz2 * vJ[4] + 2.f * cam.fx * P.tx * rz3 * vJ[2] +
                 2.f * cam.fy * P.ty * rz3 * vJ[5];
        if (v_depths) vp[2] += v_depths[o];

        // --- p = R m + t
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            vt[r] = vp[r];
#pragma unroll
            for (int k = 0; k < 3; ++k) vR[3 * r + k] = vp[r] * m[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) vm[k] = cam.R[k] * vp[0] + cam.R[3 + k] * vp[1] + cam.R[6 + k] * vp[2];

        // --- Sc = R S3 R^T:  v_R += vSc R S3^T + vSc^T R S3 ; v_S3 = R^T vSc R
        float RS3t[9];  // R * S3^T  (S3 symmetric, but keep the general form)
        mm3_abt(cam.R, S3, RS3t);
        float T1[9], T2[9];
        mm3(vSc, RS3t, T1);  // vSc (R S3^T)
        mm3_atb(vSc, RS, T2);  // vSc^T (R S3)
#pragma unroll
        for (int k = 0; k < 9; ++k) vR[k] += T1[k] + T2[k];
        float RtV[9], vS3[9];
        mm3_atb(cam.R, vSc, RtV);
        mm3(RtV, cam.R, vS3);

        // --- S3 = M M^T: v_M = (vS3 + vS3^T) M
        float Sy[9], vM[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) Sy[3 * a + b] = vS3[3 * a + b] + vS3[3 * b + a];
        mm3(Sy, M, vM);
        // --- M = Rq diag(s)
        float vRq[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) vRq[3 * r + k] = vM[3 * r + k] * s[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) vs[k] = Rq[k] * vM[k] + Rq[3 + k] * vM[3 + k] + Rq[6 + k] * vM[6 + k];
        // --- Rq = rot(normalize(q))
        const float inv = 1.f / sqrtf(qraw[1] * qraw[1] + qraw[2] * qraw[2] + qraw[3] * qraw[3] + qraw[0] * qraw[0]);
        const float w = qraw[0] * inv, x = qraw[1] * inv, y = qraw[2] * inv, z = qraw[3] * inv;
        float vn[4];
        vn[0] = 2.f * (-z * vRq[1] + y * vRq[2] + z * vRq[3] - x * vRq[5] - y * vRq[6] + x * vRq[7]);
        vn[1] = 2.f * (y * vRq[1] + z * vRq[2] + y * vRq[3] - 2.f * x * vRq[4] - w * vRq[5] + z * vRq[6] + w * vRq[7] -
                       2.f * x * vRq[8]);
        vn[2] = 2.f * (-2.f * y * vRq[0] + x * vRq[1] + w * vRq[2] + x * vRq[3] + z * vRq[5] - w * vRq[6] + z * vRq[7] -
                       2.f * y * vRq[8]);
        vn[3] = 2.f * (-2.f * z * vRq[0] - w * vRq[1] + x * vRq[2] + w * vRq[3] - 2.f * z * vRq[4] + y * vRq[5] +
                       x * vRq[6] + y * vRq[7]);
        const float dotp = vn[0] * w + vn[1] * x + vn[2] * y + vn[3] * z;
        vq[0] = (vn[0] - dotp * w) * inv;
        vq[1] = (vn[1] - dotp * x) * inv;
        vq[2] = (vn[2] - dotp * y) * inv;
        vq[3] = (vn[3] - dotp * z) * inv;
    }
    if constexpr (PREPB) {
        if (i < N) {
            float vsx[3], vo = 0.f, vc[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) vc[k] = pb.v_colors ? pb.v_colors[9 * i + k] : 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (pb.x_means) vm[k] = vm[k] + pb.x_means[3 * i + k];
                const float t = pb.x_scales ? vs[k] + pb.x_scales[3 * i + k] : vs[k];
                vsx[k] = t * scales[3 * i + k];  // d exp = exp
            }
            if (pb.x_quats) {
                const float4 r = reinterpret_cast<const float4*>(pb.x_quats)[i];
                vq[0] = vq[0] + r.x; vq[1] = vq[1] + r.y; vq[2] = vq[2] + r.z; vq[3] = vq[3] + r.w;
            }
            if (pb.v_opac) {
                const float op = pb.opac[i];
                vo = pb.v_opac[i] * op * (1.f - op);
            }
            const LeafGrads& g = pb.g;
            if (pb.accumulate)
                prep_bwd_apply<true, float>(i, pb.Ns, pb.times, pb.d_ncp, pb.d_trbf, vm, vq, vsx, vo, vc, g.s_xyz,
                                            g.s_scaling, g.s_rotation, g.s_opacity, g.s_fdc, g.s_ft, g.d_control,
                                            g.d_scaling, g.d_rotation, g.d_omega, g.d_opacity, g.d_fdc, g.d_ft);
            else
                prep_bwd_apply<false, float>(i, pb.Ns, pb.times, pb.d_ncp, pb.d_trbf, vm, vq, vsx, vo, vc, g.s_xyz,
                                             g.s_scaling, g.s_rotation, g.s_opacity, g.s_fdc, g.s_ft, g.d_control,
                                             g.d_scaling, g.d_rotation, g.d_omega, g.d_opacity, g.d_fdc, g.d_ft);
        }
    } else if (i < N) {
        if (accumulate) {
            // cameras c>0 run as later launches on the same stream, so a plain read-modify-write is safe
            v_means[3 * i] += vm[0];
            v_means[3 * i + 1] += vm[1];
            v_means[3 * i + 2] += vm[2];
            float4 a = reinterpret_cast<float4*>(v_quats)[i];
            a.x += vq[0];
            a.y += vq[1];
            a.z += vq[2];
            a.w += vq[3];
            reinterpret_cast<float4*>(v_quats)[i] = a;
        } else {
            v_means[3 * i] = vm[0];
            v_means[3 * i + 1] = vm[1];
            v_means[3 * i + 2] = vm[2];
            reinterpret_cast<float4*>(v_quats)[i] = make_float4(vq[0], vq[1], vq[2], vq[3]);
        }
        if (accumulate_scales) {  // (geometry_per_camera: positions / rotations per camera, scales shared)
            v_scales[3 * i] += vs[0];
            v_scales[3 * i + 1] += vs[1];
            v_scales[3 * i + 2] += vs[2];
        } else {
            v_scales[3 * i] = vs[0];
            v_scales[3 * i + 1] = vs[1];
            v_scales[3 * i + 2] = vs[2];
        }
    }
    // camera gradient: wave reduce -> LDS -> one 16-float partial row per workgroup (deterministic)
    if constexpr (PREPB) {
        if (!v_view_partial) return;  // (uniform over the launch: nobody asked for the pose gradient)
    }
    __shared__ float red[4][12];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float sres = wave_sum(vR[k]);
        if (lane == 0) red[wv][k] = sres;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float sres = wave_sum(vt[k]);
        if (lane == 0) red[wv][9 + k] = sres;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        // layout of one partial row: the 4x4 viewmat gradient, row-major
        const int r = threadIdx.x >> 2, k = threadIdx.x & 3;
        float v = 0.f;
        if (r < 3) {
            const int src = (k < 3) ? (3 * r + k) : (9 + r);
            v = red[0][src] + red[1][src] + red[2][src] + red[3][src];
        }
        v_view_partial[((size_t)c * gridDim.x + blockIdx.x) * 16 + threadIdx.x] = v;
    }
}

// out[i] = sum over the C per-camera copies part[c][i], in camera order (n floats)
__global__ void __launch_bounds__(256) camera_sum_kernel(int C, size_t n, const float* __restrict__ part,
                                                          float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = part[i];
    for (int c = 1; c < C; ++c) acc += part[(size_t)c * n + i];
    out[i] = acc;
}

// sums the per-workgroup partial rows: one workgroup per (camera, component), fixed order -> deterministic
__global__ void __launch_bounds__(256) viewmat_reduce_kernel(int nblocks, const float* __restrict__ partial,
                                                               float* __restrict__ v_viewmats) {
    const int c = blockIdx.x >> 4, comp = blockIdx.x & 15;
    float acc = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 256) acc += partial[((size_t)c * nblocks + b) * 16 + comp];
    acc = wave_sum(acc);
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) v_viewmats[16 * c + comp] = sm[0] + sm[1] + sm[2] + sm[3];
}

}  // namespace mobgs

using namespace mobgs;

extern "C" {

const char* mobgs_version(void) { return "mobgs_hip 0.2 gfx950"; }
int mobgs_abi_version(void) { return MOBGS_ABI_VERSION; }
const char* mobgs_last_error(void) { return g_err; }
int mobgs_record_stride(int channels) { return record_stride(channels); }

int mobgs_project_fwd(int C, int N, const float* means, const float* quats, const float* scales,
                      const float* viewmats, const float* Ks, int width, int height, float eps2d,
                      float near_plane, float far_plane, float radius_clip, int32_t* radii, float* means2d,
                      float* depths, float* conics, int32_t* tiles_per_gauss, void* stream) {
    return mobgs::project_fwd_launch(C, N, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                                     radius_clip, radii, means2d, depths, conics, tiles_per_gauss, nullptr, 0,
                                     PackArgs{nullptr, nullptr, nullptr, 0, 0, 0, 0}, stream);
}

}  // extern "C"

int mobgs::project_fwd_launch(int C, int N, const float* means, const float* quats, const float* scales,
                              const float* viewmats, const float* Ks, int width, int height, float eps2d,
                              float near_plane, float far_plane, float radius_clip, int32_t* radii, float* means2d,
                              float* depths, float* conics, int32_t* tiles_per_gauss, int32_t* zero_ptr, size_t zero_n,
                              PackArgs pack, void* stream, int geometry_per_camera, BinArgs bin,
                              const MobgsPrepInputs* prep) {
    if (C <= 0 || N < 0 || width <= 0 || height <= 0) {
        set_error("mobgs_project_fwd: bad sizes C=%d N=%d W=%d H=%d", C, N, width, height);
        return MOBGS_E_INVALID;
    }
    if (N == 0) {
        if (zero_ptr && zero_n) hipMemsetAsync(zero_ptr, 0, sizeof(int32_t) * zero_n, (hipStream_t)stream);
        return MOBGS_OK;
    }
    const int tile_w = (width + MOBGS_TILE - 1) / MOBGS_TILE, tile_h = (height + MOBGS_TILE - 1) / MOBGS_TILE;
    constexpr int T = MOBGS_PROJ_FWD_THREADS;
    dim3 grid((N + T - 1) / T, C);
    if (prep) {
        // the state is built in-kernel: means / quats / scales / pack.opacities are OUTPUT arrays here
        if (C != 1 || geometry_per_camera || prep->Ns + prep->Nd != N || !pack.records || pack.channels != 9 ||
            !pack.opacities) {
            set_error("mobgs_prep_project_and_bin_fused: one camera, Ns + Nd = N and 9-channel packed records required");
            return MOBGS_E_INVALID;
        }
        PrepFused pf;
        pf.in = PrepIn<float>{prep->Ns, prep->Nd, prep->times, prep->s_xyz, prep->s_scaling, prep->s_rotation,
                              prep->s_opacity, prep->s_fdc, prep->s_ft, prep->d_control,
                              (const long long*)prep->d_ncp, prep->d_scaling, prep->d_rotation, prep->d_omega,
                              prep->d_opacity, prep->d_fdc, prep->d_ft, prep->d_trbf};
        pf.means = const_cast<float*>(means);
        pf.quats = const_cast<float*>(quats);
        pf.scales = const_cast<float*>(scales);
        pf.opac = const_cast<float*>(pack.opacities);
        hipLaunchKernelGGL(project_fwd_kernel<true>, grid, dim3(T), 0, (hipStream_t)stream, N, means, quats, scales,
                           viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip, tile_w, tile_h,
                           radii, means2d, depths, conics, tiles_per_gauss, zero_ptr, (unsigned)zero_n, pack, 0, bin, pf);
        return check_launch("project_fwd_kernel<prep>");
    }
    hipLaunchKernelGGL(project_fwd_kernel<false>, grid, dim3(T), 0, (hipStream_t)stream, N, means, quats, scales,
                       viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip, tile_w, tile_h,
                       radii, means2d, depths, conics, tiles_per_gauss, zero_ptr, (unsigned)zero_n, pack,
                       geometry_per_camera ? N : 0, bin, PrepFused{});
    return check_launch("project_fwd_kernel");
}

extern "C" {

// [C, blocks, 16] camera-gradient partials, then (C > 1) [C, N, 3] per-camera copies of the scales' gradient
size_t mobgs_project_bwd_scratch_floats(int C, int N) {
    return (size_t)C * ((N + 255) / 256) * 16 + (C > 1 ? (size_t)C * N * 3 : 0);
}

int mobgs_project_bwd(int C, int N, const float* means, const float* quats, const float* scales,
                      const float* viewmats, const float* Ks, int width, int height, float eps2d,
                      const int32_t* radii, const float* conics, const float* v_means2d, const float* v_depths,
                      const float* v_conics, float* v_means, float* v_quats, float* v_scales,
                      float* v_viewmats, float* v_viewmats_partial, void* stream) {
    return mobgs_project_bwd_ex(C, N, 0, means, quats, scales, viewmats, Ks, width, height, eps2d, radii, conics,
                                v_means2d, v_depths, v_conics, v_means, v_quats, v_scales, v_viewmats,
                                v_viewmats_partial, stream);
}

int mobgs_project_bwd_ex(int C, int N, int geometry_per_camera, const float* means, const float* quats,
                         const float* scales, const float* viewmats, const float* Ks, int width, int height,
                         float eps2d, const int32_t* radii, const float* conics, const float* v_means2d,
                         const float* v_depths, const float* v_conics, float* v_means, float* v_quats,
                         float* v_scales, float* v_viewmats, float* v_viewmats_partial, void* stream) {
    const size_t gs = geometry_per_camera ? (size_t)N : 0;  // rows of means / quats (and their gradients) per camera
    if (C <= 0 || N < 0) {
        set_error("mobgs_project_bwd: bad sizes C=%d N=%d", C, N);
        return MOBGS_E_INVALID;
    }
    if (N == 0) {
        hipMemsetAsync(v_viewmats, 0, sizeof(float) * 16 * C, (hipStream_t)stream);
        return check_launch("project_bwd memset");
    }
    const int nblocks = (N + 255) / 256;
    if (geometry_per_camera && C > 1) {
        // ONE launch for all cameras (the K sub-frames of a blurry view: 8 launches of ~8 us at 30 k splats otherwise):
        // only the scales' gradient is shared -- every camera writes its own copy, summed in camera order (the order of
        // the sequential accumulation below: same bits)
        float* scales_partial = v_viewmats_partial + (size_t)C * nblocks * 16;
        hipLaunchKernelGGL(project_bwd_kernel<false>, dim3(nblocks, C), dim3(256), 0, (hipStream_t)stream, N, means, quats, scales,
                           viewmats, Ks, width, height, eps2d, radii, conics, v_means2d, v_depths, v_conics, v_means,
                           v_quats, scales_partial, v_viewmats_partial, 0, 0, (size_t)N, (size_t)N, PrepBwdFused{});
        hipLaunchKernelGGL(camera_sum_kernel, dim3((3 * N + 255) / 256), dim3(256), 0, (hipStream_t)stream, C,
                           (size_t)3 * N, scales_partial, v_scales);
        hipLaunchKernelGGL(viewmat_reduce_kernel, dim3(16 * C), dim3(256), 0, (hipStream_t)stream, nblocks,
                           v_viewmats_partial, v_viewmats);
        return check_launch("project_bwd_kernel");
    }
    // one launch per camera so that the accumulation into v_means/v_quats/v_scales is race-free and ordered
    for (int c = 0; c < C; ++c) {
        hipLaunchKernelGGL(project_bwd_kernel<false>, dim3(nblocks, 1), dim3(256), 0, (hipStream_t)stream, N,
                           means + 3 * c * gs, quats + 4 * c * gs, scales, viewmats + 16 * c, Ks + 9 * c, width, height,
                           eps2d, radii + (size_t)c * N, conics + (size_t)3 * c * N,
                           v_means2d ? v_means2d + (size_t)2 * c * N : nullptr,
                           v_depths ? v_depths + (size_t)c * N : nullptr,
                           v_conics ? v_conics + (size_t)3 * c * N : nullptr, v_means + 3 * c * gs, v_quats + 4 * c * gs,
                           v_scales, v_viewmats_partial + (size_t)c * nblocks * 16,
                           (c > 0 && !geometry_per_camera) ? 1 : 0, c > 0 ? 1 : 0, (size_t)0, (size_t)0, PrepBwdFused{});
    }
    hipLaunchKernelGGL(viewmat_reduce_kernel, dim3(16 * C), dim3(256), 0, (hipStream_t)stream, nblocks,
                       v_viewmats_partial, v_viewmats);
    return check_launch("project_bwd_kernel");
}

int mobgs_project_prep_bwd_fused(int N, const float* means, const float* quats, const float* scales,
                                 const float* viewmats, const float* Ks, int width, int height, float eps2d,
                                 const int32_t* radii, const float* conics, const float* v_means2d,
                                 const float* v_depths, const float* v_conics, const float* x_means,
                                 const float* x_quats, const float* x_scales, float* v_viewmats,
                                 float* v_viewmats_partial, int Ns, int Nd, const float* times, const int64_t* d_ncp,
                                 const float* d_trbf, const float* opacities, const float* v_opacities,
                                 const float* v_colors, const MobgsLeafGrads* grads, int accumulate, void* stream) {
    if (N < 1 || Ns < 0 || Nd < 0 || Ns + Nd != N || !grads || (v_viewmats && !v_viewmats_partial)) {
        set_error("mobgs_project_prep_bwd_fused: bad arguments (N=%d Ns=%d Nd=%d)", N, Ns, Nd);
        return MOBGS_E_INVALID;
    }
    PrepBwdFused pb;
    pb.Ns = Ns;
    pb.Nd = Nd;
    pb.accumulate = accumulate ? 1 : 0;
    pb.times = times;
    pb.d_ncp = (const long long*)d_ncp;
    pb.d_trbf = d_trbf;
    pb.opac = opacities;
    pb.v_opac = v_opacities;
    pb.v_colors = v_colors;
    pb.x_means = x_means;
    pb.x_quats = x_quats;
    pb.x_scales = x_scales;
    pb.g = LeafGrads{grads->s_xyz, grads->s_scaling, grads->s_rotation, grads->s_opacity, grads->s_fdc, grads->s_ft,
                     grads->d_control, grads->d_scaling, grads->d_rotation, grads->d_omega, grads->d_opacity,
                     grads->d_fdc, grads->d_ft};
    const int nblocks = (N + 255) / 256;
    hipLaunchKernelGGL(project_bwd_kernel<true>, dim3(nblocks, 1), dim3(256), 0, (hipStream_t)stream, N, means, quats,
                       scales, viewmats, Ks, width, height, eps2d, radii, conics, v_means2d, v_depths, v_conics,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, v_viewmats ? v_viewmats_partial : nullptr, 0, 0,
                       (size_t)0, (size_t)0, pb);
    if (v_viewmats)  // (NULL: the caller's camera pose does not require a gradient -- train.py never optimises it)
        hipLaunchKernelGGL(viewmat_reduce_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, nblocks,
                           v_viewmats_partial, v_viewmats);
    return check_launch("project_bwd_kernel<prep>");
}

}  // extern "C"
