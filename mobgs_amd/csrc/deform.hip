// HexPlane feature gather + deformation MLP for gfx950 (the API the reference exposes as `deform_network`).
//
// Restates for the GPU:
//   /root/reference/scene/hexplane.py:19-21,75-108,165-187   normalize_aabb, 3 levels x 6 planes bilinear
//                                                            grid_sample(align_corners=True, border), product
//                                                            over planes, concat over levels -> 96 features
//   /root/reference/scene/deformation.py:56-73,158-199       Linear(96,128); three heads ReLU-Linear(128,128)-
//                                                            ReLU-Linear(128,{7,3,4}); point / scale / rotation update
//   /root/reference/scene/deformation.py:417-438             quat2mat on [1 | dx[3:7]] / 5-norm
//   /root/reference/utils/graphics_utils.py:117-140          batch_quaternion_multiply (normalised)
//
// hexplane_fwd : 32 lanes = the 32 channels of one tap (planes are passed channels-LAST, so a bilinear tap is one
//                128-byte row); 2 points per wave instruction; the 35 MB of planes live in L2 / Infinity Cache.
// deform_mlp   : the only dense contraction on the path -> MFMA.  One wave owns 32 points through the whole
//                network (no workgroup barrier): v_mfma_f32_32x32x2_f32 (exact fp32, the vector-rate MFMA) with the
//                activation tile in a per-wave padded LDS slab and K-major weights streamed from L2.
//                63 232 MAC per point = 1152 MFMAs per 32 points.
#include <atomic>

#include "common.h"
#include "hexplane.h"

namespace mobgs {

__global__ void __launch_bounds__(256)
hexplane_fwd_kernel(int N, const float* __restrict__ pts, const float* __restrict__ times,
                    const float* __restrict__ aabb, PlaneSet planes, float* __restrict__ feat) {
    const int c = threadIdx.x & 31;
    const int half = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int nhalf = (gridDim.x * blockDim.x) >> 5;
    for (int n = half; n < N; n += nhalf) {
        float q[4], dq[3];
        normalized_query(pts, times, aabb, n, q, dq);
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            float prod = 1.f;
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const int id = l * 6 + p;
                const Tap t = make_tap(q[kAxisA[p]], q[kAxisB[p]], planes.ra[id], planes.rb[id]);
                const float* g = planes.p[id] + c;
                const float v00 = g[t.o00], v01 = g[t.o01], v10 = g[t.o10], v11 = g[t.o11];
                const float s = (v00 * (1.f - t.wx) + v01 * t.wx) * (1.f - t.wy) +
                                (v10 * (1.f - t.wx) + v11 * t.wx) * t.wy;
                prod *= s;
            }
            feat[(size_t)n * 96 + l * 32 + c] = prod;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// MLP + update rules, MFMA
// ---------------------------------------------------------------------------------------------------
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int LDA = 129;  // padded row stride: (row*129 + k) % 32 = (row + k) % 32 -> conflict-free A reads

__device__ inline void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// acc[cb] += A(32 x K, LDS, optional ReLU) * Wt(K x NCOL, global, K-major), NCB = NCOL/32 column blocks
template <int K, int NCB, bool RELU>
__device__ __forceinline__ void tile_gemm(const float (*A)[LDA], const float* __restrict__ Wt, int ncol, int lane,
                                          f32x16 (&acc)[NCB]) {
    const int r = lane & 31, kh = lane >> 5;
#pragma unroll 4
    for (int k0 = 0; k0 < K; k0 += 2) {
        float a = A[r][k0 + kh];
        if (RELU) a = fmaxf(a, 0.f);
        const float* w = Wt + (size_t)(k0 + kh) * ncol + r;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[cb * 32], acc[cb], 0, 0, 0);
    }
}

__global__ void __launch_bounds__(256)
deform_mlp_fwd_kernel(int N, const float* __restrict__ feat, const float* __restrict__ pts,
                      const float* __restrict__ scales, const float* __restrict__ rots,
                      const float* __restrict__ W0t, const float* __restrict__ b0, const float* __restrict__ W1t,
                      const float* __restrict__ b1, const float* __restrict__ W2t, const float* __restrict__ b2,
                      float* __restrict__ out_pts, float* __restrict__ out_scales, float* __restrict__ out_rots,
                      float* __restrict__ o_raw) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float(*sA)[LDA] = reinterpret_cast<float(*)[LDA]>(lds + (size_t)wv * (2 * 32 * LDA + 32 * 16));
    float(*sH)[LDA] = sA + 32;
    float(*sO)[16] = reinterpret_cast<float(*)[16]>(&sH[32][0]);
    const int row0 = (blockIdx.x * 4 + wv) * 32;
    if (row0 >= N) return;

    // feature tile -> LDS
    for (int idx = lane; idx < 32 * 96; idx += 64) {
        const int r = idx / 96, k = idx - r * 96;
        sA[r][k] = (row0 + r < N) ? feat[(size_t)(row0 + r) * 96 + k] : 0.f;
    }
    wave_fence();

    const int col = lane & 31, rbase = 4 * (lane >> 5);
    // hidden = feat W0^T + b0
    {
        f32x16 acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
        tile_gemm<96, 4, false>(sA, W0t, 128, lane, acc);
        wave_fence();
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const float bias = b0[cb * 32 + col];
#pragma unroll
            for (int i = 0; i < 16; ++i) sA[(i & 3) + 8 * (i >> 2) + rbase][cb * 32 + col] = acc[cb][i] + bias;
        }
        wave_fence();
    }
    // three heads
    const int nout[3] = {7, 3, 4};
    const int ooff[3] = {0, 7, 10};
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        {
            f32x16 acc[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
            tile_gemm<128, 4, true>(sA, W1t + (size_t)h * 128 * 128, 128, lane, acc);
            wave_fence();
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const float bias = b1[h * 128 + cb * 32 + col];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    sH[(i & 3) + 8 * (i >> 2) + rbase][cb * 32 + col] = fmaxf(acc[cb][i] + bias, 0.f);
            }
            wave_fence();
        }
        f32x16 acc2[1];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[0][i] = 0.f;
        tile_gemm<128, 1, false>(sH, W2t + (size_t)h * 128 * 32, 32, lane, acc2);
        if (col < nout[h]) {
            const float bias = b2[h * 32 + col];
#pragma unroll
            for (int i = 0; i < 16; ++i) sO[(i & 3) + 8 * (i >> 2) + rbase][ooff[h] + col] = acc2[0][i] + bias;
        }
        wave_fence();
    }
    // update rules, one lane per point
    if (lane < 32 && row0 + lane < N) {
        const int n = row0 + lane;
        float o[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) o[k] = sO[lane][k];
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
}

constexpr size_t MLP_LDS_BYTES = 4 * (2 * 32 * LDA + 32 * 16) * sizeof(float);  // 140 288 B of the 160 KiB

}  // namespace mobgs

using namespace mobgs;

extern "C" {

static int fill_planes(PlaneSet& ps, const float* const* planes_host, const int32_t* ra_host,
                       const int32_t* rb_host) {
    for (int i = 0; i < 18; ++i) {
        ps.p[i] = planes_host[i];
        ps.ra[i] = ra_host[i];
        ps.rb[i] = rb_host[i];
        if (!ps.p[i] || ps.ra[i] < 1 || ps.rb[i] < 1) return MOBGS_E_INVALID;
    }
    return MOBGS_OK;
}

int mobgs_hexplane_fwd(int N, const float* pts, const float* times, const float* aabb,
                       const float* const* planes_host, const int32_t* ra_host, const int32_t* rb_host, float* feat,
                       void* stream) {
    PlaneSet ps;
    if (N < 0 || fill_planes(ps, planes_host, ra_host, rb_host) != MOBGS_OK) {
        set_error("mobgs_hexplane_fwd: bad arguments");
        return MOBGS_E_INVALID;
    }
    if (N == 0) return MOBGS_OK;
    int grid = (N + 7) / 8;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(hexplane_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, pts, times, aabb, ps,
                       feat);
    return check_launch("hexplane_fwd_kernel");
}

int mobgs_deform_mlp_fwd(int N, const float* feat, const float* pts, const float* scales, const float* rots,
                         const float* W0t, const float* b0, const float* W1t, const float* b1, const float* W2t,
                         const float* b2, float* out_pts, float* out_scales, float* out_rots, float* o_raw,
                         void* stream) {
    if (N < 0) {
        set_error("mobgs_deform_mlp_fwd: bad N=%d", N);
        return MOBGS_E_INVALID;
    }
    if (N == 0) return MOBGS_OK;
    {  // hipFuncSetAttribute is per device: one flag per device ordinal
        static std::atomic<unsigned long long> done{0};
        int dev = 0;
        hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(deform_mlp_fwd_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)MLP_LDS_BYTES);
            done.fetch_or(bit, std::memory_order_release);
        }
    }
    const int grid = (N + 127) / 128;
    hipLaunchKernelGGL(deform_mlp_fwd_kernel, dim3(grid), dim3(256), MLP_LDS_BYTES, (hipStream_t)stream, N, feat, pts,
                       scales, rots, W0t, b0, W1t, b1, W2t, b2, out_pts, out_scales, out_rots, o_raw);
    return check_launch("deform_mlp_fwd_kernel");
}

}  // extern "C"
