// The deformation MLP + update rules (BASELINE config #3) for gfx950 on fp32 MFMA: backward pass (below), and the
// forward pass in the same workgroup tiling (deform_mlp_fwd_kernel, further down).
//
// Differentiates what deform.hip's deform_mlp_fwd_kernel computes, i.e.
//   /root/reference/scene/deformation.py:56-73,158-199   Linear(96,128); three heads ReLU-Linear(128,128)-ReLU-
//                                                        Linear(128,{7,3,4}); point / scale / rotation update
//   /root/reference/scene/deformation.py:417-438         quat2mat on [1 | dx[3:7]] / 5-norm
//   /root/reference/utils/graphics_utils.py:117-140      batch_quaternion_multiply (normalised)
// which the reference back-propagates with torch autograd (rocBLAS/cuBLAS GEMMs + ~40 elementwise kernels).
//
// One workgroup (8 waves, two per SIMD, 1 workgroup per CU: 140 KB of LDS) walks tiles of 64 points:
//   recompute   hidden -> relu -> per head z1 -> relu          (the forward keeps only the 14 raw head outputs)
//   data grads  v_o -> v_a2 -> v_z1 -> v_a1 (summed over heads in the accumulators) -> v_hidden -> v_feat
//   weight grads  gW[n][k] = sum_p V[p][n] X[p][k] as MFMAs whose reduction dimension is the POINT index: the
//               activations sit in LDS point-major, so both operands of a v_mfma_f32_32x32x2_f32 are plain
//               conflict-free row reads; the 64 accumulator tiles (W0, 3 x W1, W2) stay in registers for the
//               whole walk (wave w owns output rows 32 (w & 3) .. +31 and one half of the columns of every matrix:
//               160 accumulator registers per lane) and are written ONCE per workgroup as a partial; mlp_grad_reduce_kernel sums the partials in workgroup order
//               (deterministic: no float atomics).
// All GEMMs are v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation (bitwise an fmaf chain), the vector
// fp32 rate (157 TFLOP/s on MI355X) without spending VALU issue slots on it.
#include <atomic>

#include "common.h"

namespace mobgs {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int TP = 64;     // points per workgroup tile
constexpr int LDB = 129;   // padded row stride of the LDS slabs (column reads of 32 different rows hit 32 banks)
constexpr int LDO = 33;
#ifndef GEMM_U
#define GEMM_U 4
#endif    // row stride of the head-output cotangents

// partial / final gradient block (floats)
constexpr int OFF_W0 = 0;                      // [128][96]
constexpr int OFF_B0 = OFF_W0 + 128 * 96;      // [128]
constexpr int OFF_W1 = OFF_B0 + 128;           // [3][128][128]
constexpr int OFF_B1 = OFF_W1 + 3 * 128 * 128; // [3][128]
constexpr int OFF_W2 = OFF_B1 + 3 * 128;       // [32][128]  rows 8h..8h+7 = head h (pos 7, scl 3, rot 4 used)
constexpr int OFF_B2 = OFF_W2 + 32 * 128;      // [32]
constexpr int GRAD_FLOATS = OFF_B2 + 32;       // 66 080

struct BwdLds {
    float F[TP][LDB];    // HexPlane features of the tile (96 columns used)
    float A1[TP][LDB];   // relu(hidden)
    float A2[TP][LDB];   // relu(z1) of the current head
    float VZ[TP][LDB];   // v_z1 of the current head; v_hidden at the end
    float VO[TP][LDO];   // head-output cotangents, column 8h + o
};

__device__ inline void zero(f32x16& a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
}

// row of accumulator element i of a 32x32 tile held by `lane`
__device__ __forceinline__ int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// acc += A[row0 + ., 0..K) * B[0..K)[col0 + .]   (A: LDS slab, B: global, row stride ldb)
// Operands are fetched one group of U k-steps AHEAD of the MFMAs that consume them (explicit double buffering in
// registers): a dependent v_mfma_f32_32x32x2_f32 chain issues one instruction per 64 cycles, so without the
// look-ahead the loads of a group only leave once the previous group's last MFMA has issued and their latency
// (LDS ~100, L2 ~300-500 cycles) is exposed every U MFMAs.
template <int K>
__device__ __forceinline__ void gemm_tile(const float (*A)[LDB], int row0, const float* __restrict__ B, int ldb,
                                          int col0, int lane, f32x16& acc, int acol0 = 0) {
    constexpr int U = GEMM_U;      // k-steps per group
    constexpr int G = K / 2 / U;   // groups; processed two at a time (ping / pong register sets)
    static_assert(G % 2 == 0, "K/2 must be a multiple of twice the group size");
    const int r = lane & 31, kh = lane >> 5;
    const float* ap = &A[row0 + r][acol0 + kh];
    const float* bp = B + (size_t)kh * ldb + col0 + r;
    float a0[U], b0[U], a1[U], b1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        a0[u] = ap[2 * u];
        b0[u] = bp[(size_t)(2 * u) * ldb];
    }
#pragma unroll 1
    for (int g = 0; g < G; g += 2) {
        {
            const float* an = ap + 2 * U * (g + 1);
            const float* bn = bp + (size_t)(2 * U * (g + 1)) * ldb;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a1[u] = an[2 * u];
                b1[u] = bn[(size_t)(2 * u) * ldb];
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the loads AHEAD of the MFMAs (the scheduler would sink them)
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 < G) {
            const float* an = ap + 2 * U * (g + 2);
            const float* bn = bp + (size_t)(2 * U * (g + 2)) * ldb;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a0[u] = an[2 * u];
                b0[u] = bn[(size_t)(2 * u) * ldb];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Update rules backward for one point: raw head outputs o[14], inputs, output cotangents -> vo[14] and the
// cotangents of the (pts, rots) inputs (the scales input cotangent is v_out_scales itself).
__device__ inline void update_rules_bwd(const float* o, const float p_in[3], const float r_in[4], const float g[3],
                                        const float gs[3], const float gr[4], float vo[14], float v_p_in[3],
                                        float v_r_in[4]) {
    // ---- points: out = R(q) P, P = p_in + o[0:3], q = (1, o3, o4, o5) / sqrt(1 + o3^2 + o4^2 + o5^2 + o6^2)
    const float P[3] = {p_in[0] + o[0], p_in[1] + o[1], p_in[2] + o[2]};
    const float s5 = 1.f + o[3] * o[3] + o[4] * o[4] + o[5] * o[5] + o[6] * o[6];
    const float inv5 = 1.f / sqrtf(s5);
    const float w = inv5, x = o[3] * inv5, y = o[4] * inv5, z = o[5] * inv5;
    const float R[9] = {w * w + x * x - y * y - z * z, 2.f * x * y - 2.f * w * z, 2.f * w * y + 2.f * x * z,
                        2.f * w * z + 2.f * x * y, w * w - x * x + y * y - z * z, 2.f * y * z - 2.f * w * x,
                        2.f * x * z - 2.f * w * y, 2.f * w * x + 2.f * y * z, w * w - x * x - y * y + z * z};
    float vP[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) vP[j] = R[j] * g[0] + R[3 + j] * g[1] + R[6 + j] * g[2];
    float G[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) G[3 * i + j] = g[i] * P[j];
    const float vw = 2.f * (w * (G[0] + G[4] + G[8]) - z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7]);
    const float vx = 2.f * (x * (G[0] - G[4] - G[8]) + y * G[1] + z * G[2] + y * G[3] - w * G[5] + z * G[6] + w * G[7]);
    const float vy = 2.f * (y * (-G[0] + G[4] - G[8]) + x * G[1] + w * G[2] + x * G[3] + z * G[5] - w * G[6] + z * G[7]);
    const float vz = 2.f * (z * (-G[0] - G[4] + G[8]) - w * G[1] + x * G[2] + w * G[3] + y * G[5] + x * G[6] + y * G[7]);
    const float v_inv5 = vw + vx * o[3] + vy * o[4] + vz * o[5];
    const float v_s5 = -0.5f * v_inv5 * inv5 * inv5 * inv5;
    vo[0] = vP[0];
    vo[1] = vP[1];
    vo[2] = vP[2];
    vo[3] = vx * inv5 + 2.f * o[3] * v_s5;
    vo[4] = vy * inv5 + 2.f * o[4] * v_s5;
    vo[5] = vz * inv5 + 2.f * o[5] * v_s5;
    vo[6] = 2.f * o[6] * v_s5;
    v_p_in[0] = vP[0];
    v_p_in[1] = vP[1];
    v_p_in[2] = vP[2];
    // ---- scales: out = s_in + clamp(o[7:10], +-log 100)   (torch.clamp passes the gradient on [min, max])
    const float L = 4.605170185988092f;
#pragma unroll
    for (int k = 0; k < 3; ++k) vo[7 + k] = (o[7 + k] >= -L && o[7 + k] <= L) ? gs[k] : 0.f;
    // ---- rotations: out = normalize((r_in + o[10:14]) (x) o[3:7])
    const float a0 = r_in[0] + o[10], a1 = r_in[1] + o[11], a2 = r_in[2] + o[12], a3 = r_in[3] + o[13];
    const float b0 = o[3], b1 = o[4], b2 = o[5], b3 = o[6];
    const float qw = a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3;
    const float qx = a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2;
    const float qy = a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1;
    const float qz = a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0;
    const float invn = 1.f / sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
    const float nw = qw * invn, nx = qx * invn, ny = qy * invn, nz = qz * invn;
    const float dot = nw * gr[0] + nx * gr[1] + ny * gr[2] + nz * gr[3];
    const float vqw = (gr[0] - nw * dot) * invn, vqx = (gr[1] - nx * dot) * invn, vqy = (gr[2] - ny * dot) * invn,
                vqz = (gr[3] - nz * dot) * invn;
    const float va0 = vqw * b0 + vqx * b1 + vqy * b2 + vqz * b3;
    const float va1 = -vqw * b1 + vqx * b0 - vqy * b3 + vqz * b2;
    const float va2 = -vqw * b2 + vqx * b3 + vqy * b0 - vqz * b1;
    const float va3 = -vqw * b3 - vqx * b2 + vqy * b1 + vqz * b0;
    vo[10] = va0;
    vo[11] = va1;
    vo[12] = va2;
    vo[13] = va3;
    v_r_in[0] = va0;
    v_r_in[1] = va1;
    v_r_in[2] = va2;
    v_r_in[3] = va3;
    vo[3] += vqw * a0 + vqx * a1 + vqy * a2 + vqz * a3;
    vo[4] += -vqw * a1 + vqx * a0 + vqy * a3 - vqz * a2;
    vo[5] += -vqw * a2 - vqx * a3 + vqy * a0 + vqz * a1;
    vo[6] += -vqw * a3 + vqx * a2 - vqy * a1 + vqz * a0;
}

// One thread per point: cotangents of the three outputs -> cotangents of the 14 raw head outputs (v_o [N,16]) and
// of the point / rotation inputs.
__global__ void __launch_bounds__(256)
update_rules_bwd_kernel(int N, const float* __restrict__ pts, const float* __restrict__ rots,
                        const float* __restrict__ o_raw, const float* __restrict__ v_out_pts,
                        const float* __restrict__ v_out_scales, const float* __restrict__ v_out_rots,
                        float* __restrict__ v_o, float* __restrict__ v_pts, float* __restrict__ v_rots) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float o[16], vo[16], p_in[3], r_in[4], g[3], gs[3], gr[4], vp[3], vr[4];
    const float4* src = reinterpret_cast<const float4*>(o_raw + (size_t)n * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t = src[q];
        o[4 * q] = t.x, o[4 * q + 1] = t.y, o[4 * q + 2] = t.z, o[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p_in[k] = pts[3 * n + k];
        g[k] = v_out_pts ? v_out_pts[3 * n + k] : 0.f;
        gs[k] = v_out_scales ? v_out_scales[3 * n + k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r_in[k] = rots[4 * n + k];
        gr[k] = v_out_rots ? v_out_rots[4 * n + k] : 0.f;
    }
    vo[14] = vo[15] = 0.f;
    update_rules_bwd(o, p_in, r_in, g, gs, gr, vo, vp, vr);
    float4* dst = reinterpret_cast<float4*>(v_o + (size_t)n * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_float4(vo[4 * q], vo[4 * q + 1], vo[4 * q + 2], vo[4 * q + 3]);
#pragma unroll
    for (int k = 0; k < 3; ++k) v_pts[3 * n + k] = vp[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) v_rots[4 * n + k] = vr[k];
}

struct MlpWeights {
    const float* W0t;  // [96][128]   feature_out weight, K-major (transposed)
    const float* b0;   // [128]
    const float* W1t;  // [3][128][128] K-major
    const float* b1;   // [3][128]
    const float* W0;   // [128][96]   original (out, in) layouts for the data gradients
    const float* W1;   // [3][128][128]
    const float* W2;   // [3][8][128] rows >= nout are zero
};

__global__ void __launch_bounds__(512)
deform_mlp_bwd_kernel(int N, int n_tiles, const float* __restrict__ feat, const float* __restrict__ v_o,
                      MlpWeights Wt, float* __restrict__ v_feat, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    BwdLds& S = *reinterpret_cast<BwdLds*>(lds_raw);
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int r = lane & 31, kh = lane >> 5;
    // data GEMMs [64 points x 128]: this wave's output tile = rows 32 hf .., columns col0 ..
    // weight gradients [128 x 128]: rows col0 .. (the same 32-wide block), column tiles 2 hf, 2 hf + 1
    const int col0_ = 32 * (wv & 3), hf = wv >> 2, prow0_ = 32 * hf;

    // gW1 lives in registers for the whole walk.  The W0 and W2 gradient tiles are touched in one short phase per
    // tile each: they are PARKED in this workgroup's partial block between tiles (48 + 16 KB, L2-resident; every
    // lane re-reads exactly the addresses it wrote) -- with them resident the kernel needs ~300 registers per lane
    // and spills inside the MFMA loops.
    f32x16 gW1[3][2];
    float gb1[3] = {0.f, 0.f, 0.f}, gb0 = 0.f, gb2 = 0.f;
    float* const P = partials + (size_t)blockIdx.x * GRAD_FLOATS;
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        zero(gW1[h][0]);
        zero(gW1[h][1]);
    }

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * TP;
        // An opaque zero folded into the per-wave offsets: the compiler would otherwise hoist every address that
        // does not depend on the tile out of this loop (3 unrolled heads x 6 arrays) and spill them (149 VGPRs).
        int zv = 0;
        asm volatile("" : "+v"(zv));
        const int col0 = col0_ + zv, prow0 = prow0_ + zv;
        // ---- P0: features and head-output cotangents -> LDS ---------------------------------------------
        for (int idx = tid; idx < TP * 96; idx += 512) {
            const int p = idx / 96, k = idx - p * 96;
            S.F[p][k] = (row0 + p < N) ? feat[(size_t)(row0 + p) * 96 + k] : 0.f;
        }
        // head-output cotangents (update_rules_bwd_kernel): columns 0..6 | 7..9 | 10..13 -> 8h + o
        for (int idx = tid; idx < TP * 32; idx += 512) {
            const int p = idx >> 5, c = idx & 31;
            const int h = c >> 3, o = c & 7;
            const int src = (h == 0) ? (o < 7 ? o : -1) : (h == 1) ? (o < 3 ? 7 + o : -1) : (h == 2) ? (o < 4 ? 10 + o : -1) : -1;
            S.VO[p][c] = (src >= 0 && row0 + p < N) ? v_o[(size_t)(row0 + p) * 16 + src] : 0.f;
        }
        __syncthreads();
        // ---- P1: hidden = F W0^T + b0 -> A1 = relu(hidden) -----------------------------------------------
        {
            f32x16 acc;
            zero(acc);
            gemm_tile<96>(S.F, prow0, Wt.W0t, 128, col0, lane, acc);
            const float bias = Wt.b0[col0 + r];
#pragma unroll
            for (int i = 0; i < 16; ++i) S.A1[prow0 + acc_row(i, lane)][col0 + r] = fmaxf(acc[i] + bias, 0.f);
        }
        __syncthreads();
        f32x16 va1, gW2;
        zero(va1);
        zero(gW2);
        const bool first = tile == (int)blockIdx.x;
        if (hf == 1 && !first) {
#pragma unroll
            for (int i = 0; i < 16; ++i) gW2[i] = P[OFF_W2 + acc_row(i, lane) * 128 + col0 + r];
        }
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            // ---- P2: z1 = A1 W1^T + b1 -> A2 = relu(z1) ---------------------------------------------------
            {
                f32x16 acc;
                zero(acc);
                gemm_tile<128>(S.A1, prow0, Wt.W1t + (size_t)h * 128 * 128, 128, col0, lane, acc);
                const float bias = Wt.b1[h * 128 + col0 + r];
#pragma unroll
                for (int i = 0; i < 16; ++i) S.A2[prow0 + acc_row(i, lane)][col0 + r] = fmaxf(acc[i] + bias, 0.f);
            }
            // ---- P3: v_a2 = VO_h W2 -> VZ = v_a2 where z1 > 0 (each lane reads back only what it wrote) ----
            {
                f32x16 acc;
                zero(acc);
                const float* W2h = Wt.W2 + (size_t)h * 8 * 128;
#pragma unroll
                for (int k0 = 0; k0 < 8; k0 += 2) {
                    const int kk = k0 + kh;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(S.VO[prow0 + r][8 * h + kk], W2h[kk * 128 + col0 + r],
                                                               acc, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = prow0 + acc_row(i, lane);
                    S.VZ[row][col0 + r] = (S.A2[row][col0 + r] > 0.f) ? acc[i] : 0.f;
                }
            }
            __syncthreads();
            // ---- P4: weight gradients of this head (reduction over the tile's 64 points) + v_a1 ------------
            {
                float bsum = 0.f, osum = 0.f;
                const bool mine = (r >> 3) == h;  // VO column r belongs to head h
#pragma unroll 4
                for (int p0 = 0; p0 < TP; p0 += 2) {
                    const int p = p0 + kh;
                    const float a = S.VZ[p][col0 + r];  // A[m = n][k = p]
                    bsum += a;
                    gW1[h][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, S.A1[p][64 * hf + r], gW1[h][0], 0, 0, 0);
                    gW1[h][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, S.A1[p][64 * hf + 32 + r], gW1[h][1], 0, 0, 0);
                    if (hf == 1) {  // wave-uniform: the upper four waves also carry the W2 gradient
                        const float vo = mine ? S.VO[p][r] : 0.f;  // A[m = o][k = p], other heads' columns masked
                        osum += vo;
                        gW2 = __builtin_amdgcn_mfma_f32_32x32x2f32(vo, S.A2[p][col0 + r], gW2, 0, 0, 0);
                    }
                }
                gb1[h] += bsum;
                gb2 += osum;
                // v_a1 += VZ W1 (original layout: row n, column k)
                gemm_tile<128>(S.VZ, prow0, Wt.W1 + (size_t)h * 128 * 128, 128, col0, lane, va1);
            }
            __syncthreads();
        }
        if (hf == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) P[OFF_W2 + acc_row(i, lane) * 128 + col0 + r] = gW2[i];
        }
        // ---- P5: v_hidden = v_a1 where hidden > 0 -> VZ slab ---------------------------------------------
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = prow0 + acc_row(i, lane);
            S.VZ[row][col0 + r] = (S.A1[row][col0 + r] > 0.f) ? va1[i] : 0.f;
        }
        __syncthreads();
        // ---- P6: gW0 += VH^T F ; gb0 ; v_feat = VH W0 ---------------------------------------------------
        {
            f32x16 gW0[2];
            zero(gW0[0]);
            zero(gW0[1]);
            if (!first) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = col0 + acc_row(i, lane);
                    gW0[0][i] = P[OFF_W0 + row * 96 + 64 * hf + r];
                    if (hf == 0) gW0[1][i] = P[OFF_W0 + row * 96 + 32 + r];
                }
            }
            float bsum = 0.f;
#pragma unroll 4
            for (int p0 = 0; p0 < TP; p0 += 2) {
                const int p = p0 + kh;
                const float a = S.VZ[p][col0 + r];
                bsum += a;
                // column tiles of gW0 [128 x 96]: lower waves take 0 and 1, upper waves take 2
                gW0[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, S.F[p][64 * hf + r], gW0[0], 0, 0, 0);
                if (hf == 0) gW0[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, S.F[p][32 + r], gW0[1], 0, 0, 0);
            }
            gb0 += bsum;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = col0 + acc_row(i, lane);
                P[OFF_W0 + row * 96 + 64 * hf + r] = gW0[0][i];
                if (hf == 0) P[OFF_W0 + row * 96 + 32 + r] = gW0[1][i];
            }
            if (col0 < 96) {  // 2 x 3 output tiles of v_feat [64 x 96]
                f32x16 acc;
                zero(acc);
                gemm_tile<128>(S.VZ, prow0, Wt.W0, 96, col0, lane, acc);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int n = row0 + prow0 + acc_row(i, lane);
                    if (n < N) v_feat[(size_t)n * 96 + col0 + r] = acc[i];
                }
            }
        }
        __syncthreads();
    }

    // ---- this workgroup's partial sums (W0 / W2 are already in place) -----------------------------------------
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = col0_ + acc_row(i, lane);
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            P[OFF_W1 + h * 16384 + row * 128 + 64 * hf + r] = gW1[h][0][i];
            P[OFF_W1 + h * 16384 + row * 128 + 64 * hf + 32 + r] = gW1[h][1][i];
        }
    }
    // bias sums: lanes l and l + 32 hold the two halves of the point sum
    gb0 += __shfl_xor(gb0, 32, 64);
    gb2 += __shfl_xor(gb2, 32, 64);
#pragma unroll
    for (int h = 0; h < 3; ++h) gb1[h] += __shfl_xor(gb1[h], 32, 64);
    if (kh == 0 && hf == 1) {
        P[OFF_B0 + col0_ + r] = gb0;
#pragma unroll
        for (int h = 0; h < 3; ++h) P[OFF_B1 + h * 128 + col0_ + r] = gb1[h];
        if (wv == 4) P[OFF_B2 + r] = gb2;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward: the same workgroup tiling (8 waves, 64 points, pipelined operand fetch) instead of round 1's one-wave-per-
// 32-points kernel (31 % of the fp32 MFMA peak: every MFMA waited for its own L2 fetch).  The 128 -> {7,3,4} output
// layers have only 64 x 32 outputs per tile: the eight waves split their K = 128 into quarters (wave (cb, hf): rows of
// half hf, k in [32 cb, 32 cb + 32)), leave the four partial tiles in LDS and the workgroup sums them in fixed order.
struct FwdLds {
    float F[TP][LDB];        // features; reused for the partial output tiles of the heads
    float A1[TP][LDB];       // relu(hidden)
    float A2[TP][LDB];       // relu(z1) of the current head
    float PO[4][TP][LDO];    // partial head outputs per K-quarter
    float O[TP][16];         // the 14 raw head outputs
};

__global__ void __launch_bounds__(512)
deform_mlp_fwd_kernel(int N, int n_tiles, const float* __restrict__ feat, const float* __restrict__ pts,
                      const float* __restrict__ scales, const float* __restrict__ rots, const float* __restrict__ W0t,
                      const float* __restrict__ b0, const float* __restrict__ W1t, const float* __restrict__ b1,
                      const float* __restrict__ W2t, const float* __restrict__ b2, float* __restrict__ out_pts,
                      float* __restrict__ out_scales, float* __restrict__ out_rots, float* __restrict__ o_raw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    FwdLds& S = *reinterpret_cast<FwdLds*>(lds_raw);
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int r = lane & 31;
    const int cb = wv & 3, hf = wv >> 2, col0 = 32 * cb, prow0 = 32 * hf;
    const int nout[3] = {7, 3, 4}, ooff[3] = {0, 7, 10};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * TP;
        for (int idx = tid; idx < TP * 96; idx += 512) {
            const int p = idx / 96, k = idx - p * 96;
            S.F[p][k] = (row0 + p < N) ? feat[(size_t)(row0 + p) * 96 + k] : 0.f;
        }
        __syncthreads();
        {
            f32x16 acc;
            zero(acc);
            gemm_tile<96>(S.F, prow0, W0t, 128, col0, lane, acc);
            const float bias = b0[col0 + r];
#pragma unroll
            for (int i = 0; i < 16; ++i) S.A1[prow0 + acc_row(i, lane)][col0 + r] = fmaxf(acc[i] + bias, 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            {
                f32x16 acc;
                zero(acc);
                gemm_tile<128>(S.A1, prow0, W1t + (size_t)h * 128 * 128, 128, col0, lane, acc);
                const float bias = b1[h * 128 + col0 + r];
#pragma unroll
                for (int i = 0; i < 16; ++i) S.A2[prow0 + acc_row(i, lane)][col0 + r] = fmaxf(acc[i] + bias, 0.f);
            }
            __syncthreads();
            {
                f32x16 acc;
                zero(acc);
                gemm_tile<32>(S.A2, prow0, W2t + (size_t)h * 128 * 32 + (size_t)col0 * 32, 32, 0, lane, acc, col0);
#pragma unroll
                for (int i = 0; i < 16; ++i) S.PO[cb][prow0 + acc_row(i, lane)][r] = acc[i];
            }
            __syncthreads();
            for (int idx = tid; idx < TP * 8; idx += 512) {
                const int p = idx >> 3, o = idx & 7;
                if (o < nout[h])
                    S.O[p][ooff[h] + o] = ((S.PO[0][p][o] + S.PO[1][p][o]) + (S.PO[2][p][o] + S.PO[3][p][o])) +
                                          b2[h * 32 + o];
            }
            // (the next head's first barrier orders these reads of PO / writes of O before anything overwrites them)
        }
        __syncthreads();
        // update rules, one lane per point
        if (tid < TP && row0 + tid < N) {
            const int n = row0 + tid;
            float o[14];
#pragma unroll
            for (int k = 0; k < 14; ++k) o[k] = S.O[tid][k];
            if (o_raw) {  // what the backward pass needs of the forward: the 14 raw head outputs (64 B per point)
#pragma unroll
                for (int k = 0; k < 14; ++k) o_raw[(size_t)n * 16 + k] = o[k];
            }
            // points: R(quat2mat5(dx[3:7])) (p + dx[0:3])
            const float px = pts[3 * n] + o[0], py = pts[3 * n + 1] + o[1], pz = pts[3 * n + 2] + o[2];
            const float inv5 = 1.f / sqrtf(1.f + o[3] * o[3] + o[4] * o[4] + o[5] * o[5] + o[6] * o[6]);
            const float w = inv5, x = o[3] * inv5, y = o[4] * inv5, z = o[5] * inv5;
            const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
            const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
            out_pts[3 * n] = (w2 + x2 - y2 - z2) * px + (2.f * xy - 2.f * wz) * py + (2.f * wy + 2.f * xz) * pz;
            out_pts[3 * n + 1] = (2.f * wz + 2.f * xy) * px + (w2 - x2 + y2 - z2) * py + (2.f * yz - 2.f * wx) * pz;
            out_pts[3 * n + 2] = (2.f * xz - 2.f * wy) * px + (2.f * wx + 2.f * yz) * py + (w2 - x2 - y2 + z2) * pz;
            // scales: + clamp(ds, +-log 100)
            const float L = 4.605170185988092f;
#pragma unroll
            for (int k = 0; k < 3; ++k) out_scales[3 * n + k] = scales[3 * n + k] + fminf(fmaxf(o[7 + k], -L), L);
            // rotations: normalize((rot + dr) (x) dx[3:7])
            const float a0 = rots[4 * n] + o[10], a1 = rots[4 * n + 1] + o[11], a2 = rots[4 * n + 2] + o[12],
                        a3 = rots[4 * n + 3] + o[13];
            const float b0q = o[3], b1q = o[4], b2q = o[5], b3q = o[6];
            const float qw = a0 * b0q - a1 * b1q - a2 * b2q - a3 * b3q;
            const float qx = a0 * b1q + a1 * b0q + a2 * b3q - a3 * b2q;
            const float qy = a0 * b2q - a1 * b3q + a2 * b0q + a3 * b1q;
            const float qz = a0 * b3q + a1 * b2q - a2 * b1q + a3 * b0q;
            const float invn = 1.f / sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
            out_rots[4 * n] = qw * invn;
            out_rots[4 * n + 1] = qx * invn;
            out_rots[4 * n + 2] = qy * invn;
            out_rots[4 * n + 3] = qz * invn;
        }
        __syncthreads();
    }
}

// out[i] = sum over the workgroups' partials, in workgroup order
__global__ void __launch_bounds__(256) mlp_grad_reduce_kernel(int n_part, const float* __restrict__ partials,
                                                              float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= GRAD_FLOATS) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 4 <= n_part; b += 4) {
        s0 += partials[(size_t)b * GRAD_FLOATS + i];
        s1 += partials[(size_t)(b + 1) * GRAD_FLOATS + i];
        s2 += partials[(size_t)(b + 2) * GRAD_FLOATS + i];
        s3 += partials[(size_t)(b + 3) * GRAD_FLOATS + i];
    }
    for (; b < n_part; ++b) s0 += partials[(size_t)b * GRAD_FLOATS + i];
    out[i] = (s0 + s1) + (s2 + s3);
}

// hipFuncSetAttribute is per device: one flag per (kernel, device ordinal)
static void allow_dynamic_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
}

}  // namespace mobgs

using namespace mobgs;

extern "C" {

int mobgs_deform_mlp_fwd(int N, const float* feat, const float* pts, const float* scales, const float* rots,
                         const float* W0t, const float* b0, const float* W1t, const float* b1, const float* W2t,
                         const float* b2, float* out_pts, float* out_scales, float* out_rots, float* o_raw,
                         void* stream) {
    if (N < 0) {
        set_error("mobgs_deform_mlp_fwd: bad N=%d", N);
        return MOBGS_E_INVALID;
    }
    if (N == 0) return MOBGS_OK;
    static std::atomic<unsigned long long> attr_done{0};
    allow_dynamic_lds(reinterpret_cast<const void*>(deform_mlp_fwd_kernel), (int)sizeof(FwdLds), attr_done);
    const int tiles = (N + TP - 1) / TP;
    const int grid = tiles < 256 ? tiles : 256;
    hipLaunchKernelGGL(deform_mlp_fwd_kernel, dim3(grid), dim3(512), sizeof(FwdLds), (hipStream_t)stream, N, tiles, feat,
                       pts, scales, rots, W0t, b0, W1t, b1, W2t, b2, out_pts, out_scales, out_rots, o_raw);
    return check_launch("deform_mlp_fwd_kernel");
}

int mobgs_deform_mlp_bwd_blocks(int N) {
    const int tiles = (N + TP - 1) / TP;
    return tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256;
}

size_t mobgs_deform_mlp_grad_floats(void) { return (size_t)GRAD_FLOATS; }

int mobgs_deform_mlp_bwd(int N, const float* feat, const float* pts, const float* rots, const float* o_raw,
                         const float* W0t, const float* b0, const float* W1t, const float* b1, const float* W0,
                         const float* W1, const float* W2pad, const float* v_out_pts, const float* v_out_scales,
                         const float* v_out_rots, float* v_feat, float* v_pts, float* v_rots, float* v_o,
                         float* partials, float* grads, void* stream) {
    if (N < 0 || !grads ||
        (N > 0 && (!feat || !pts || !rots || !o_raw || !partials || !v_feat || !v_pts || !v_rots || !v_o))) {
        set_error("mobgs_deform_mlp_bwd: bad arguments (N=%d)", N);
        return MOBGS_E_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        hipMemsetAsync(grads, 0, GRAD_FLOATS * sizeof(float), s);
        return MOBGS_OK;
    }
    static std::atomic<unsigned long long> attr_done{0};
    allow_dynamic_lds(reinterpret_cast<const void*>(deform_mlp_bwd_kernel), (int)sizeof(BwdLds), attr_done);
    const int tiles = (N + TP - 1) / TP;
    const int grid = mobgs_deform_mlp_bwd_blocks(N);
    MlpWeights Wt{W0t, b0, W1t, b1, W0, W1, W2pad};
    hipLaunchKernelGGL(update_rules_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, pts, rots, o_raw, v_out_pts,
                       v_out_scales, v_out_rots, v_o, v_pts, v_rots);
    int rc = check_launch("update_rules_bwd_kernel");
    if (rc != MOBGS_OK) return rc;
    hipLaunchKernelGGL(deform_mlp_bwd_kernel, dim3(grid), dim3(512), sizeof(BwdLds), s, N, tiles, feat, v_o, Wt, v_feat,
                       partials);
    rc = check_launch("deform_mlp_bwd_kernel");
    if (rc != MOBGS_OK) return rc;
    hipLaunchKernelGGL(mlp_grad_reduce_kernel, dim3((GRAD_FLOATS + 255) / 256), dim3(256), 0, s, grid, partials, grads);
    return check_launch("mlp_grad_reduce_kernel");
}

}  // extern "C"
