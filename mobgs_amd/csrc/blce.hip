// BLCE (blur-aware latent camera + exposure estimation) of ONE view, forward and backward, as two single-wave kernels.
//
// Restates /root/reference/scene/blce.py:374-478 (BLCE.forward: blur-feature embedding + encoder, Rt encoder, view
// encoder, 8 explicit Euler steps of WV_Derivative (:234-275, torchdiffeq 'euler' on the integer grid, :278-309),
// rotation / translation / angle decoders, SE(3) exponential, pose composition) and the pose inversion of
// blceKernel.get_warped_cams (:150-152, torch.inverse).  The reference runs this as ~180 forward and ~350 backward
// torch launches of a few microseconds of work each per view: pure launch latency (1.9 / 5.9 ms eagerly; 0.46 / 2.6 ms
// replayed as a HIP graph, round 1).  The whole thing is ~40 k multiply-adds on 32-wide vectors: one wave64 with the
// vectors in LDS does it in one launch each way.  In the sub-frame-sharded training step every rank repeats BLCE for
// each view it renders a sub-frame of, so this cost is NOT divided by the number of GPUs -- at 8 GPUs it would exceed
// the rendering time per rank.
//
// Parameter table (22 device pointers, HOST array), the reference's module tree for view `idx`:
//   0 view_embedder [num_views,32]      1,2  Rt_encoder.{weight [32,12], bias}     3,4  view_encoder.{weight [32,64], bias}
//   5..10 blur_feature_encoder.{0,2,4}.{weight,bias}  ([32,21],[32] | [32,32],[32] | [32,32],[32])
//   11 wv_derivative.time_embedder [9,8]  12,13 w_linear.{weight [16,56], bias}  14,15 v_linear.{weight [16,56], bias}
//   16,17 rot_decoder.{weight [3,16], bias}  18,19 trans_decoder.{weight [3,16], bias}  20,21 theta_decoder.{weight [1,16], bias}
// The gradient table has the same order and shapes (entry 0: the full [num_views,32] table, only row idx is written;
// every gradient tensor is FULLY written).
#include "common.h"

namespace mobgs {

constexpr int NW = 9;      // latent sub-frames (num_warp)
constexpr int VD = 32;     // view_dim
constexpr int NF = 10;     // frequencies of the blur-feature embedding
constexpr int EMB = 2 * NF + 1;
// saved activations (floats): what the backward kernel re-reads
constexpr int S_EMBED = 0;                 // [21]
constexpr int S_H1 = S_EMBED + 24;         // [32]
constexpr int S_H2 = S_H1 + VD;            // [32]
constexpr int S_E = S_H2 + VD;             // [32] encoded blur feature
constexpr int S_VIEW = S_E + VD;           // [64] cat(view_embedder[idx], Rt_encoder(Rt))
constexpr int S_X = S_VIEW + 64;           // [9][32] latent trajectory
constexpr int S_DEC = S_X + NW * VD;       // [9][8]  rot(3) theta(1) trans(3) pad
constexpr int S_M = S_DEC + NW * 8;        // [9][16] warped c2w
constexpr int S_W = S_M + NW * 16;         // [9][16] warped w2c
constexpr int S_TOTAL = S_W + NW * 16;     // 848 floats

struct BlceParams {
    const float* p[22];
};
struct BlceGrads {
    float* p[22];
};

// out[o] = b[o] + sum_k W[o][k] x[k], o < n_out (lane = output row); x in LDS
__device__ __forceinline__ float matvec(const float* __restrict__ W, const float* __restrict__ b, const float* x,
                                        int n_in, int o, int n_out) {
    if (o >= n_out) return 0.f;
    float acc = b ? b[o] : 0.f;
    const float* w = W + (size_t)o * n_in;
    for (int k = 0; k < n_in; ++k) acc = __fmaf_rn(w[k], x[k], acc);
    return acc;
}
// y[k] = sum_o W[o][k] v[o], k < n_in (lane = input column): the transposed product of the backward pass
__device__ __forceinline__ float matvec_t(const float* __restrict__ W, const float* v, int n_in, int n_out, int k) {
    if (k >= n_in) return 0.f;
    float acc = 0.f;
    for (int o = 0; o < n_out; ++o) acc = __fmaf_rn(W[(size_t)o * n_in + k], v[o], acc);
    return acc;
}

__device__ inline void inverse4(const float* m, float* inv) {
    // general 4x4 inverse by cofactors (what torch.inverse returns up to rounding; the poses are rigid)
    float c[16];
    c[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    c[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    c[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    c[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    c[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    c[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    c[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    c[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    c[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    c[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    c[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    c[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    c[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    c[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    c[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    c[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * c[0] + m[1] * c[4] + m[2] * c[8] + m[3] * c[12];
    const float id = 1.0f / det;
#pragma unroll
    for (int i = 0; i < 16; ++i) inv[i] = c[i] * id;
}

struct Se3 {
    float K[9], K2[9], R[9], G[9], p[3], s, c, n;
};
// rot [3], theta, trans [3] -> K, K^2, R = exp, G, p = G trans   (scene/blce.py:432-470)
__device__ inline Se3 se3_exp(const float* rot, float th, const float* tr) {
    Se3 e;
    e.n = sqrtf(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2]);
    const float inv = 1.f / (e.n + 1e-10f);
    const float u1 = rot[0] * inv, u2 = rot[1] * inv, u3 = rot[2] * inv;
    const float K[9] = {0.f, -u3, u2, u3, 0.f, -u1, -u2, u1, 0.f};
#pragma unroll
    for (int i = 0; i < 9; ++i) e.K[i] = K[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) e.K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
    e.s = sinf(th);
    e.c = cosf(th);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float I = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
        e.R[i] = I + e.s * e.K[i] + (1.f - e.c) * e.K2[i];
        e.G[i] = I * th + (1.f - e.c) * e.K[i] + (th - e.s) * e.K2[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) e.p[i] = e.G[3 * i] * tr[0] + e.G[3 * i + 1] * tr[1] + e.G[3 * i + 2] * tr[2];
    return e;
}

__global__ void __launch_bounds__(64)
blce_fwd_kernel(BlceParams P, int idx, const float* __restrict__ Rt, const float* __restrict__ bf_ptr,
                float* __restrict__ c2w_out, float* __restrict__ w2c_out, float* __restrict__ saved) {
    __shared__ float sv[S_TOTAL];
    __shared__ float win[2][56];
    const int t = threadIdx.x;
    const float bf = bf_ptr[0];
    // blur-feature embedding [bf, sin(bf 2^i pi), cos(bf 2^i pi)]
    if (t == 0) sv[S_EMBED] = bf;
    if (t < NF) {
        const float a = bf * (float)(1 << t) * 3.14159265358979323846f;
        sv[S_EMBED + 1 + t] = sinf(a);
        sv[S_EMBED + 1 + NF + t] = cosf(a);
    }
    __syncthreads();
    const float h1 = fmaxf(matvec(P.p[5], P.p[6], sv + S_EMBED, EMB, t, VD), 0.f);
    if (t < VD) sv[S_H1 + t] = h1;
    __syncthreads();
    const float h2 = fmaxf(matvec(P.p[7], P.p[8], sv + S_H1, VD, t, VD), 0.f);
    if (t < VD) sv[S_H2 + t] = h2;
    __syncthreads();
    const float e = matvec(P.p[9], P.p[10], sv + S_H2, VD, t, VD);
    if (t < VD) sv[S_E + t] = e;
    // view = [view_embedder[idx] | Rt_encoder(Rt[:3,:])]
    if (t < VD) sv[S_VIEW + t] = P.p[0][(size_t)idx * VD + t];
    {
        float acc = 0.f;
        if (t < VD) {
            acc = P.p[2][t];
            for (int k = 0; k < 12; ++k) acc = __fmaf_rn(P.p[1][t * 12 + k], Rt[k], acc);  // Rt[:3,:] = first 12 entries
            sv[S_VIEW + VD + t] = acc;
        }
    }
    __syncthreads();
    const float x0 = matvec(P.p[3], P.p[4], sv + S_VIEW, 64, t, VD);
    if (t < VD) sv[S_X + t] = x0;
    __syncthreads();
    // explicit Euler, dt = 1: x_{i+1} = x_i + [w_linear([relu(x)[:16], temb_i, e]) ; v_linear([relu(x)[16:], temb_i, e])]
    for (int i = 0; i < NW - 1; ++i) {
        if (t < 16) {
            win[0][t] = fmaxf(sv[S_X + i * VD + t], 0.f);
            win[1][t] = fmaxf(sv[S_X + i * VD + 16 + t], 0.f);
        } else if (t < 24) {
            win[0][t] = win[1][t] = P.p[11][i * 8 + (t - 16)];
        } else if (t < 56) {
            win[0][t] = win[1][t] = sv[S_E + (t - 24)];
        }
        __syncthreads();
        float d = 0.f;
        if (t < 16) d = matvec(P.p[12], P.p[13], win[0], 56, t, 16);
        else if (t < 32) d = matvec(P.p[14], P.p[15], win[1], 56, t - 16, 16);
        if (t < VD) sv[S_X + (i + 1) * VD + t] = sv[S_X + i * VD + t] + d;
        __syncthreads();
    }
    // decoders + SE(3) exponential + pose composition + inverse: lane j < 9 handles sub-frame j
    if (t < NW) {
        const float* lw = sv + S_X + t * VD;
        const float* lv = lw + 16;
        float rot[3], tr[3], th;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rot[k] = matvec(P.p[16], P.p[17], lw, 16, k, 3);
            tr[k] = matvec(P.p[18], P.p[19], lv, 16, k, 3);
        }
        th = matvec(P.p[20], P.p[21], lw, 16, 0, 1);
        float* dec = sv + S_DEC + t * 8;
        dec[0] = rot[0], dec[1] = rot[1], dec[2] = rot[2], dec[3] = th, dec[4] = tr[0], dec[5] = tr[1], dec[6] = tr[2], dec[7] = 0.f;
        const Se3 E = se3_exp(rot, th, tr);
        float T[16] = {E.R[0], E.R[1], E.R[2], E.p[0], E.R[3], E.R[4], E.R[5], E.p[1],
                       E.R[6], E.R[7], E.R[8], E.p[2], 0.f, 0.f, 0.f, 1.f};
        float M[16], Wi[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                M[4 * i + j] = Rt[4 * i] * T[j] + Rt[4 * i + 1] * T[4 + j] + Rt[4 * i + 2] * T[8 + j] + Rt[4 * i + 3] * T[12 + j];
        inverse4(M, Wi);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sv[S_M + t * 16 + i] = M[i];
            sv[S_W + t * 16 + i] = Wi[i];
            c2w_out[t * 16 + i] = M[i];
            w2c_out[t * 16 + i] = Wi[i];
        }
    }
    __syncthreads();
    if (saved)
        for (int i = t; i < S_TOTAL; i += 64) saved[i] = sv[i];
}

// gW[o][k] = v[o] * x[k] for o < n_out (lane = k), += when ACC
__device__ __forceinline__ void outer(float* __restrict__ gW, const float* v, const float* x, int n_out, int n_in, int k,
                                      bool acc) {
    if (k >= n_in) return;
    for (int o = 0; o < n_out; ++o) {
        const float val = v[o] * x[k];
        gW[(size_t)o * n_in + k] = acc ? gW[(size_t)o * n_in + k] + val : val;
    }
}

__global__ void __launch_bounds__(64)
blce_bwd_kernel(BlceParams P, BlceGrads Gd, int idx, int num_views, const float* __restrict__ Rt,
                const float* __restrict__ saved, const float* __restrict__ v_c2w, const float* __restrict__ v_w2c) {
    __shared__ float sv[S_TOTAL];
    __shared__ float vx[NW][VD];     // cotangents of the latent trajectory
    __shared__ float vdec[NW][8];    // cotangents of rot(3) theta(1) trans(3)
    __shared__ float ve[VD], vtmp[64], win[2][56], vin[2][56];
    const int t = threadIdx.x;
    for (int i = t; i < S_TOTAL; i += 64) sv[i] = saved[i];
    if (t < VD) ve[t] = 0.f;
    __syncthreads();
    // ---- pose inverse, composition, SE(3) exponential: lane j < 9
    if (t < NW) {
        const float* M = sv + S_M + t * 16;
        const float* Wi = sv + S_W + t * 16;
        float VM[16];
        // W = M^-1: v_M = v_c2w - W^T v_W W^T
        float A[16];  // A = W^T v_W
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) a += Wi[4 * k + i] * (v_w2c ? v_w2c[t * 16 + 4 * k + j] : 0.f);
                A[4 * i + j] = a;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) a += A[4 * i + k] * Wi[4 * j + k];
                VM[4 * i + j] = (v_c2w ? v_c2w[t * 16 + 4 * i + j] : 0.f) - a;
            }
        (void)M;
        // M = Rt T: v_T = Rt^T v_M (top three rows of T carry parameters)
        float VR[9], Vp[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) a += Rt[4 * k + i] * VM[4 * k + j];
                if (j < 3) VR[3 * i + j] = a; else Vp[i] = a;
            }
        const float* dec = sv + S_DEC + t * 8;
        const float rot[3] = {dec[0], dec[1], dec[2]}, tr[3] = {dec[4], dec[5], dec[6]};
        const float th = dec[3];
        const Se3 E = se3_exp(rot, th, tr);
        // p = G tr
        float VG[9], Vtr[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) VG[3 * i + j] = Vp[i] * tr[j];
            Vtr[i] = E.G[i] * Vp[0] + E.G[3 + i] * Vp[1] + E.G[6 + i] * Vp[2];
        }
        // R = I + s K + (1-c) K2 ; G = th I + (1-c) K + (th - s) K2
        float dRK = 0.f, dRK2 = 0.f, dGK = 0.f, dGK2 = 0.f, trG = VG[0] + VG[4] + VG[8];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            dRK += VR[i] * E.K[i];
            dRK2 += VR[i] * E.K2[i];
            dGK += VG[i] * E.K[i];
            dGK2 += VG[i] * E.K2[i];
        }
        const float Vs = dRK - dGK2, Vc = -dRK2 - dGK;
        const float Vth = trG + dGK2 + Vs * E.c - Vc * E.s;
        float VK2[9], VK[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            VK2[i] = (1.f - E.c) * VR[i] + (th - E.s) * VG[i];
            VK[i] = E.s * VR[i] + (1.f - E.c) * VG[i];
        }
        // K2 = K K: v_K += v_K2 K^T + K^T v_K2
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) a += VK2[3 * i + k] * E.K[3 * j + k] + E.K[3 * k + i] * VK2[3 * k + j];
                VK[3 * i + j] += a;
            }
        const float Vu[3] = {VK[7] - VK[5], VK[2] - VK[6], VK[3] - VK[1]};
        // u = rot / (n + eps)
        const float d = E.n + 1e-10f;
        const float dotu = Vu[0] * rot[0] + Vu[1] * rot[1] + Vu[2] * rot[2];
        const float coef = (E.n > 0.f) ? dotu / (d * d * E.n) : 0.f;
        vdec[t][0] = Vu[0] / d - rot[0] * coef;
        vdec[t][1] = Vu[1] / d - rot[1] * coef;
        vdec[t][2] = Vu[2] / d - rot[2] * coef;
        vdec[t][3] = Vth;
        vdec[t][4] = Vtr[0];
        vdec[t][5] = Vtr[1];
        vdec[t][6] = Vtr[2];
        vdec[t][7] = 0.f;
    }
    __syncthreads();
    // ---- decoders: v_x[j] = [W_rot^T v_rot + W_th^T v_th ; W_tr^T v_tr]; their weight gradients (sum over j)
    for (int j = 0; j < NW; ++j) {
        float a = 0.f;
        if (t < 16) {
            for (int o = 0; o < 3; ++o) a = __fmaf_rn(P.p[16][o * 16 + t], vdec[j][o], a);
            a = __fmaf_rn(P.p[20][t], vdec[j][3], a);
        } else if (t < 32) {
            for (int o = 0; o < 3; ++o) a = __fmaf_rn(P.p[18][o * 16 + (t - 16)], vdec[j][4 + o], a);
        }
        if (t < VD) vx[j][t] = a;
    }
    if (t < 16) {
        float gr[3] = {0.f, 0.f, 0.f}, gt[3] = {0.f, 0.f, 0.f}, gth = 0.f;
        for (int j = 0; j < NW; ++j) {
            const float lw = sv[S_X + j * VD + t], lv = sv[S_X + j * VD + 16 + t];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                gr[o] = __fmaf_rn(vdec[j][o], lw, gr[o]);
                gt[o] = __fmaf_rn(vdec[j][4 + o], lv, gt[o]);
            }
            gth = __fmaf_rn(vdec[j][3], lw, gth);
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            Gd.p[16][o * 16 + t] = gr[o];
            Gd.p[18][o * 16 + t] = gt[o];
        }
        Gd.p[20][t] = gth;
    }
    if (t < 7) {  // biases: rot 0..2, theta 3, trans 4..6
        float a = 0.f;
        for (int j = 0; j < NW; ++j) a += vdec[j][t];
        if (t < 3) Gd.p[17][t] = a;
        else if (t == 3) Gd.p[21][0] = a;
        else Gd.p[19][t - 4] = a;
    }
    __syncthreads();
    // ---- Euler steps in reverse
    for (int i = NW - 2; i >= 0; --i) {
        // inputs of step i (recomputed from x_i)
        if (t < 16) {
            win[0][t] = fmaxf(sv[S_X + i * VD + t], 0.f);
            win[1][t] = fmaxf(sv[S_X + i * VD + 16 + t], 0.f);
        } else if (t < 24) {
            win[0][t] = win[1][t] = P.p[11][i * 8 + (t - 16)];
        } else if (t < 56) {
            win[0][t] = win[1][t] = sv[S_E + (t - 24)];
        }
        __syncthreads();
        // v_dw = vx[i+1][:16], v_dv = vx[i+1][16:]
        const bool first = i == NW - 2;
        outer(Gd.p[12], &vx[i + 1][0], win[0], 16, 56, t, !first);
        outer(Gd.p[14], &vx[i + 1][16], win[1], 16, 56, t, !first);
        if (t < 16) {
            Gd.p[13][t] = (first ? 0.f : Gd.p[13][t]) + vx[i + 1][t];
            Gd.p[15][t] = (first ? 0.f : Gd.p[15][t]) + vx[i + 1][16 + t];
        }
        const float a0 = matvec_t(P.p[12], &vx[i + 1][0], 56, 16, t);
        const float a1 = matvec_t(P.p[14], &vx[i + 1][16], 56, 16, t);
        __syncthreads();
        if (t < 56) {
            vin[0][t] = a0;
            vin[1][t] = a1;
        }
        __syncthreads();
        if (t >= 16 && t < 24) Gd.p[11][i * 8 + (t - 16)] = vin[0][t] + vin[1][t];  // time_embedder row i
        if (t >= 24 && t < 56) ve[t - 24] += vin[0][t] + vin[1][t];
        if (t < VD) {
            const float xa = sv[S_X + i * VD + t];
            const float va = t < 16 ? vin[0][t] : vin[1][t - 16];
            vx[i][t] += vx[i + 1][t] + (xa > 0.f ? va : 0.f);
        }
        __syncthreads();
    }
    if (t < 8) Gd.p[11][(NW - 1) * 8 + t] = 0.f;  // the last time embedding row is never used
    // ---- view encoder: x0 = W_V view + b_V
    outer(Gd.p[3], &vx[0][0], sv + S_VIEW, VD, 64, t, false);
    if (t < VD) Gd.p[4][t] = vx[0][t];
    const float vview = matvec_t(P.p[3], &vx[0][0], 64, VD, t);
    vtmp[t] = vview;
    __syncthreads();
    // view_embedder table: only row idx
    for (int i = t; i < num_views * VD; i += 64) Gd.p[0][i] = 0.f;
    __syncthreads();
    if (t < VD) Gd.p[0][(size_t)idx * VD + t] = vtmp[t];
    // Rt encoder: r = W_R Rt12 + b_R
    if (t < 12) {
        for (int o = 0; o < VD; ++o) Gd.p[1][o * 12 + t] = vtmp[VD + o] * Rt[t];
    }
    if (t < VD) Gd.p[2][t] = vtmp[VD + t];
    // ---- blur-feature encoder: e = W_c h2 + b_c ; h2 = relu(W_b h1 + b_b) ; h1 = relu(W_a embed + b_a)
    outer(Gd.p[9], ve, sv + S_H2, VD, VD, t, false);
    if (t < VD) Gd.p[10][t] = ve[t];
    float vh2 = matvec_t(P.p[9], ve, VD, VD, t);
    if (t < VD) vh2 = sv[S_H2 + t] > 0.f ? vh2 : 0.f;
    __syncthreads();
    if (t < VD) vtmp[t] = vh2;
    __syncthreads();
    outer(Gd.p[7], vtmp, sv + S_H1, VD, VD, t, false);
    if (t < VD) Gd.p[8][t] = vtmp[t];
    float vh1 = matvec_t(P.p[7], vtmp, VD, VD, t);
    if (t < VD) vh1 = sv[S_H1 + t] > 0.f ? vh1 : 0.f;
    __syncthreads();
    if (t < VD) vtmp[32 + t] = vh1;
    __syncthreads();
    outer(Gd.p[5], vtmp + 32, sv + S_EMBED, VD, EMB, t, false);
    if (t < VD) Gd.p[6][t] = vtmp[32 + t];
}

}  // namespace mobgs

using namespace mobgs;

extern "C" {

size_t mobgs_blce_saved_floats(void) { return (size_t)S_TOTAL; }

int mobgs_blce_fwd(const float* const* params_host, int idx, int num_views, const float* Rt,
                   const float* blur_feature, float* c2w, float* w2c, float* saved, void* stream) {
    if (!params_host || !Rt || !blur_feature || !c2w || !w2c) {
        set_error("mobgs_blce_fwd: bad arguments");
        return MOBGS_E_INVALID;
    }
    if (idx < 0 || idx >= num_views) {  // view_embedder has num_views rows
        set_error("mobgs_blce_fwd: view index %d outside [0, %d)", idx, num_views);
        return MOBGS_E_INVALID;
    }
    BlceParams P;
    for (int i = 0; i < 22; ++i) {
        P.p[i] = params_host[i];
        if (!P.p[i]) {
            set_error("mobgs_blce_fwd: parameter %d is NULL", i);
            return MOBGS_E_INVALID;
        }
    }
    hipLaunchKernelGGL(blce_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, P, idx, Rt, blur_feature, c2w, w2c,
                       saved);
    return check_launch("blce_fwd_kernel");
}

int mobgs_blce_bwd(const float* const* params_host, float* const* grads_host, int idx, int num_views, const float* Rt,
                   const float* saved, const float* v_c2w, const float* v_w2c, void* stream) {
    if (!params_host || !grads_host || !Rt || !saved || idx < 0 || idx >= num_views) {
        set_error("mobgs_blce_bwd: bad arguments");
        return MOBGS_E_INVALID;
    }
    BlceParams P;
    BlceGrads G;
    for (int i = 0; i < 22; ++i) {
        P.p[i] = params_host[i];
        G.p[i] = grads_host[i];
        if (!P.p[i] || !G.p[i]) {
            set_error("mobgs_blce_bwd: parameter / gradient %d is NULL", i);
            return MOBGS_E_INVALID;
        }
    }
    hipLaunchKernelGGL(blce_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, P, G, idx, num_views, Rt, saved, v_c2w,
                       v_w2c);
    return check_launch("blce_bwd_kernel");
}

}  // extern "C"
