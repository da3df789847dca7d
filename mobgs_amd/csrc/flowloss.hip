// Fused flow-consistency loss for gfx950: the consumer of get_flow()'s four outputs in a training iteration.
//
// Restates for the GPU /root/reference/train.py:651-671 (with utils/loss_utils.py:233-237, the masked l1_loss):
//   coords  -> x / (W - 1), y / (H - 1), 2 n - 1                                   (:653-656, :661-664)
//   warp_1  =  F.grid_sample(ori_image[b] (expanded over K), exp2mid grid, bilinear, padding_mode='border')   (:658)
//   warp_2  =  F.grid_sample(latent_img[b,k],                mid2exp grid, bilinear, padding_mode='border')   (:666)
//   loss    =  l1(warp_1, latent_img, mask=latent_alpha) + l1(warp_2, ori_image, mask=d_alpha)                (:668)
//   l1(a, b, mask) = sum |(a - b) mask| / (sum(mask expanded to 3 channels) + 1e-8)
// (align_corners is torch's default, False: ix = ((g + 1) W - 1) / 2, clipped to [0, W - 1]; taps outside the image
// contribute nothing.)  lambda_flow_loss is applied by the caller.
//
// The reference spends two grid_sample calls on [B K,3,H,W] tensors (the first on a materialised K-fold copy of the
// mid image), ~20 element-wise launches over tensors of that size, and torch's grid_sampler_2d_backward.  Here one
// forward kernel reads every input once and leaves four partial sums per workgroup; one backward kernel recomputes
// the samples and writes all six gradients.  A thread owns one pixel (b, y, x) (backward: a column of FL_ROWS pixels)
// and walks the K exposures.
//
// Gradients with respect to the SAMPLED images are scatters (the transpose of the bilinear gather): float atomics, as
// in torch, so those two gradients are sums in an order that is not fixed.  The L2 retires about one float atomic per
// channel and clock, which bounds the backward kernel; two things cut their number: (i) a pixel whose mask is
// exactly zero contributes exactly zero and issues none; (ii) the flow is smooth, so the right-hand taps of lane l
// are almost always the left-hand taps of lane l + 1 and the lower taps of row y those of row y + 1: a wave walks a
// strip of rows, the right-hand column moves one lane up, the lower row is carried to the next row, and about ONE
// dense atomic instruction per (row, channel, warp) leaves instead of four (struct Scatter / StrayQueue below).
#include <hip/hip_runtime.h>

#include "common.h"

namespace mobgs {

constexpr int FL_TX = 64, FL_TY = 4, FL_THREADS = FL_TX * FL_TY;

struct Taps {
    int x0, y0;       // north-west tap (always inside the image)
    float ex, fx;     // (x0 + 1) - ix, ix - x0
    float ey, fy;
    float mx, my;     // d(clipped coordinate) / d(coordinate): 0 where the border clamps
    bool right, down; // x0 + 1 / y0 + 1 inside the image
};

// pixel coordinate c (un-normalised, as get_flow returns it) -> sample position, with the reference's arithmetic:
// n = c / (size - 1); g = 2 n - 1 (train.py); ix = ((g + 1) size - 1) / 2 (grid_sampler_unnormalize, align_corners
// False); clip to [0, size - 1] (padding_mode='border'; NaN -> 0 like fmax)
__device__ __forceinline__ float sample_pos(float c, int size, float& mult) {
    const float n = __fdiv_rn(c, (float)(size - 1));
    const float g = __fsub_rn(__fmul_rn(2.f, n), 1.f);
    float i = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(g, 1.f), (float)size), 1.f), 0.5f);
    const float hi = (float)(size - 1);
    mult = 1.f;
    if (!(i > 0.f)) {
        i = 0.f;
        mult = 0.f;
    } else if (i >= hi) {
        i = hi;
        mult = 0.f;
    }
    return i;
}

__device__ __forceinline__ Taps locate(float cx, float cy, int W, int H) {
    Taps t;
    const float ix = sample_pos(cx, W, t.mx), iy = sample_pos(cy, H, t.my);
    const float x0f = floorf(ix), y0f = floorf(iy);
    t.x0 = (int)x0f;
    t.y0 = (int)y0f;
    t.ex = (x0f + 1.f) - ix;
    t.fx = ix - x0f;
    t.ey = (y0f + 1.f) - iy;
    t.fy = iy - y0f;
    t.right = t.x0 + 1 < W;
    t.down = t.y0 + 1 < H;
    return t;
}

struct Quad {
    float v00, v01, v10, v11;  // nw, ne, sw, se (0 where the tap is outside)
};
__device__ __forceinline__ Quad fetch(const float* __restrict__ plane, const Taps& t, int W) {
    const float* p = plane + (size_t)t.y0 * W + t.x0;
    Quad q;
    q.v00 = p[0];
    q.v01 = t.right ? p[1] : 0.f;
    q.v10 = t.down ? p[W] : 0.f;
    q.v11 = (t.right && t.down) ? p[W + 1] : 0.f;
    return q;
}
__device__ __forceinline__ float bilinear(const Quad& q, const Taps& t) {
    // grid_sampler_2d: out = nw_val * nw + ne_val * ne + sw_val * sw + se_val * se, accumulated in that order
    float o = q.v00 * (t.ex * t.ey);
    o = __fmaf_rn(q.v01, t.fx * t.ey, o);
    o = __fmaf_rn(q.v10, t.ex * t.fy, o);
    o = __fmaf_rn(q.v11, t.fx * t.fy, o);
    return o;
}

__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// ori [B,3,H,W], latent [B,K,3,H,W], e2m / m2e [B,K,H,W,2], la [B,K,H,W], da [B,H,W]
// partial [n_blocks,4] = {sum |d1 la|, sum la, sum |d2 da|, sum da (per exposure)} of the workgroup's pixels
__global__ void __launch_bounds__(FL_THREADS)
flow_warp_l1_fwd_kernel(int B, int K, int H, int W, const float* __restrict__ ori, const float* __restrict__ latent,
                        const float* __restrict__ e2m, const float* __restrict__ m2e, const float* __restrict__ la,
                        const float* __restrict__ da, float* __restrict__ partial) {
    __shared__ float red[4];
    const int x = blockIdx.x * FL_TX + (threadIdx.x & (FL_TX - 1));
    const int y = blockIdx.y * FL_TY + (threadIdx.x / FL_TX);
    const int b = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const bool live = x < W && y < H;
    float n1 = 0.f, s1 = 0.f, n2 = 0.f, s2 = 0.f;
    if (live) {
        const size_t pix = (size_t)y * W + x;
        const float* ob = ori + (size_t)b * 3 * plane;
        const float o0 = ob[pix], o1 = ob[plane + pix], o2 = ob[2 * plane + pix];
        const float dav = da[(size_t)b * plane + pix];
        for (int k = 0; k < K; ++k) {
            const size_t bk = (size_t)b * K + k;
            const float2 c1 = reinterpret_cast<const float2*>(e2m)[bk * plane + pix];
            const float2 c2 = reinterpret_cast<const float2*>(m2e)[bk * plane + pix];
            const float lav = la[bk * plane + pix];
            const float* lb = latent + bk * 3 * plane;
            const Taps t1 = locate(c1.x, c1.y, W, H);
            const Taps t2 = locate(c2.x, c2.y, W, H);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float w1 = bilinear(fetch(ob + c * plane, t1, W), t1);
                const float w2 = bilinear(fetch(lb + c * plane, t2, W), t2);
                const float oc = c == 0 ? o0 : (c == 1 ? o1 : o2);
                n1 += fabsf((w1 - lb[c * plane + pix]) * lav);
                n2 += fabsf((w2 - oc) * dav);
            }
            s1 += lav;
            s2 += dav;
        }
    }
    const float r0 = block_sum(n1, red), r1 = block_sum(s1, red), r2 = block_sum(n2, red), r3 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        float* p = partial + 4 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        p[0] = r0;
        p[1] = r1;
        p[2] = r2;
        p[3] = r3;
    }
}

// one workgroup: the four sums in a fixed order (double accumulators) -> sums[4] = {N1, S1, N2, S2} with S = 3 x the
// mask sum (the mask is expanded to the three colour channels before it is summed) and
// loss = N1 / (S1 + 1e-8) + N2 / (S2 + 1e-8)
__global__ void __launch_bounds__(256) flow_warp_l1_finish_kernel(int n_blocks, const float* __restrict__ partial,
                                                                   float* __restrict__ sums, float* __restrict__ loss) {
    __shared__ double sm[4][256];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < n_blocks; i += 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += (double)partial[4 * (size_t)i + q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sm[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sm[q][threadIdx.x] += sm[q][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float N1 = (float)sm[0][0], S1 = (float)(3.0 * sm[1][0]);
        const float N2 = (float)sm[2][0], S2 = (float)(3.0 * sm[3][0]);
        sums[0] = N1;
        sums[1] = S1;
        sums[2] = N2;
        sums[3] = S2;
        loss[0] = N1 / (S1 + 1e-8f) + N2 / (S2 + 1e-8f);
    }
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// value of the lane below (lane - 1) of this wave; lane 0 gets 0
__device__ __forceinline__ float from_left(float v, int lane) {
    const float r = __shfl_up(v, 1, 64);
    return lane == 0 ? 0.f : r;
}

__device__ __forceinline__ void fl_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-wave queue of stray contributions (address of channel 0 + the three channel values).  What an atomic costs here
// is the INSTRUCTION, not the lane (measured: the backward pass's time follows the number of atomic instructions
// issued, one lane active or sixty-four): the few contributions per row that no neighbour absorbs -- the right-hand
// column of lane 63, cells where the flow breaks -- would each be an instruction with one or two live lanes.  They
// are parked in LDS instead and leave 64 at a time.
constexpr int FL_QCAP = 192;
struct StrayQueue {
    float** addr;       // LDS [FL_QCAP]
    float* val;         // LDS [FL_QCAP][3]
    int count;          // wave-uniform
    size_t plane;

    __device__ __forceinline__ void drain(int lane) {
        fl_wave_fence();
        for (int i = lane; i < count; i += 64) {
            float* a = addr[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = val[3 * i + c];
                if (v != 0.f) unsafeAtomicAdd(a + c * plane, v);
            }
        }
        count = 0;
        fl_wave_fence();
    }
    __device__ __forceinline__ void push(bool p, float* a, float v0, float v1, float v2, int lane) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(p);
        if (m == 0ull) return;
        if (count + 64 > FL_QCAP) drain(lane);
        if (p) {
            const int at = count + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            addr[at] = a;
            val[3 * at] = v0;
            val[3 * at + 1] = v1;
            val[3 * at + 2] = v2;
        }
        count += __builtin_popcountll(m);
    }
};

// The scatter of one warp's image gradient (the transpose of its bilinear gather) for a wave that walks a strip of 64
// columns row by row.  Per row every lane has up to four contributions per channel (nw, ne, sw, se of its cell).  With
// a smooth flow the cell of lane l + 1 is the right-hand neighbour of lane l's cell, and the cell of the next row is
// the one below: COMBINE moves the right-hand column one lane up and CARRIES the lower row to the next row of the
// walk, so that what leaves as (dense) atomics is about one value per (pixel, channel) instead of four; what no
// neighbour absorbs goes through the stray queue.  Same sums (up to the order of additions).
template <bool COMBINE>
struct Scatter {
    float c10[3], c11[3];  // carried: contributions to row cy, columns cx and cx + 1
    int cx, cy;
    bool have;

    __device__ __forceinline__ void init() {
        have = false;
        cx = cy = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) c10[c] = c11[c] = 0.f;
    }
    // end of the strip: the carried row leaves -- its left column densely, the right one is whatever was not given away
    __device__ __forceinline__ void flush(float* __restrict__ dst, size_t plane, int W, StrayQueue& q, int lane) {
        if (!COMBINE) return;
        float* p = dst + (size_t)cy * W + cx;
        if (have) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c10[c] != 0.f) unsafeAtomicAdd(p + c * plane, c10[c]);
        }
        q.push(have && (c11[0] != 0.f || c11[1] != 0.f || c11[2] != 0.f), p + 1, c11[0], c11[1], c11[2], lane);
        have = false;
    }
    // one row: a[c] = d loss / d (sampled value of channel c) of this lane's pixel (active: any of them non-zero)
    __device__ __forceinline__ void row(float* __restrict__ dst, size_t plane, int W, const Taps& t, bool active,
                                        const float (&a)[3], int lane, StrayQueue& q) {
        bool take = false;  // this lane receives its left neighbour's right-hand column
        bool give = false;  // this lane's right-hand column leaves with the right neighbour
        if (COMBINE) {
            const int nx0 = __shfl_down(t.x0, 1, 64), ny0 = __shfl_down(t.y0, 1, 64);
            const int nact = __shfl_down((int)active, 1, 64);
            give = active && lane < 63 && nact != 0 && nx0 == t.x0 + 1 && ny0 == t.y0;
            take = __shfl_up((int)give, 1, 64) != 0 && lane > 0;
        }
        // the carried row continues in this lane's cell iff this cell is the one below the carried one
        const bool merge = COMBINE && have && active && t.x0 == cx && t.y0 == cy;
        if (COMBINE) {  // a carried row that does not continue here: stray
            const bool stray = have && !merge;
            float* pc = dst + (size_t)cy * W + cx;
            q.push(stray && (c10[0] != 0.f || c10[1] != 0.f || c10[2] != 0.f), pc, c10[0], c10[1], c10[2], lane);
            q.push(stray && (c11[0] != 0.f || c11[1] != 0.f || c11[2] != 0.f), pc + 1, c11[0], c11[1], c11[2], lane);
            if (stray) have = false;
        }
        const float w00 = t.ex * t.ey, w01 = t.fx * t.ey, w10 = t.ex * t.fy, w11 = t.fx * t.fy;
        float* p = dst + (size_t)t.y0 * W + t.x0;
        float ne[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = active ? a[c] : 0.f;
            float v00 = g * w00, v01 = g * w01, v10 = g * w10, v11 = g * w11;
            if (merge) {
                v00 += c10[c];
                v01 += c11[c];
            }
            if (COMBINE) {
                const float l01 = from_left(give ? v01 : 0.f, lane), l11 = from_left(give ? v11 : 0.f, lane);
                if (take) {
                    v00 += l01;
                    v10 += l11;
                }
                if (give) v01 = v11 = 0.f;
            }
            float* pc = p + c * plane;
            ne[c] = (active && t.right) ? v01 : 0.f;
            if (active) {
                if (v00 != 0.f) unsafeAtomicAdd(pc, v00);
                if (COMBINE) {  // the lower row waits for the next row of the walk
                    c10[c] = t.down ? v10 : 0.f;
                    c11[c] = (t.down && t.right) ? v11 : 0.f;
                } else {
                    if (t.right && v01 != 0.f) unsafeAtomicAdd(pc + 1, v01);
                    if (t.down && v10 != 0.f) unsafeAtomicAdd(pc + W, v10);
                    if (t.down && t.right && v11 != 0.f) unsafeAtomicAdd(pc + W + 1, v11);
                }
            }
        }
        if (COMBINE) {
            q.push(ne[0] != 0.f || ne[1] != 0.f || ne[2] != 0.f, p + 1, ne[0], ne[1], ne[2], lane);
            if (active) {
                have = t.down;
                cx = t.x0;
                cy = t.y0 + 1;
            }
        }
    }
};

// d loss / d (pixel coordinate) from d loss / d (sampled values) and the tap values
__device__ __forceinline__ void coord_grad(const Quad& q, const Taps& t, float g, float& gix, float& giy) {
    gix = __fmaf_rn(g, (q.v01 - q.v00) * t.ey + (q.v11 - q.v10) * t.fy, gix);
    giy = __fmaf_rn(g, (q.v10 - q.v00) * t.ex + (q.v11 - q.v01) * t.fx, giy);
}

// FL_ROWS = rows of a wave's strip in the backward pass (a workgroup: 4 strips): 8 on large images (fewer strip ends:
// 3.97 -> 3.81 ms for two views of 1352x1014), 4 where 8 would leave CUs without a workgroup

// v_loss: device scalar (the cotangent of the loss).  Bilinear scatters are ACCUMULATED into g_ori [B,3,H,W] and
// g_latent [B,K,3,H,W] with atomics (zero-filled by the caller); the DIRECT gradients of the two images (as targets of
// the other term: one value per element) go to d_ori / d_latent of the same shapes with plain stores;
// g_e2m / g_m2e [B,K,H,W,2], g_la [B,K,H,W], g_da [B,H,W] are fully written.  The image pairs may be NULL together.
template <bool COMBINE, int FL_ROWS>
__global__ void __launch_bounds__(FL_THREADS)
flow_warp_l1_bwd_kernel(int B, int K, int H, int W, const float* __restrict__ ori, const float* __restrict__ latent,
                        const float* __restrict__ e2m, const float* __restrict__ m2e, const float* __restrict__ la,
                        const float* __restrict__ da, const float* __restrict__ sums,
                        const float* __restrict__ v_loss, float* __restrict__ g_ori, float* __restrict__ d_ori,
                        float* __restrict__ g_latent, float* __restrict__ d_latent, float* __restrict__ g_e2m,
                        float* __restrict__ g_m2e, float* __restrict__ g_la, float* __restrict__ g_da) {
    __shared__ float* q_addr[FL_THREADS / 64][FL_QCAP];
    __shared__ float q_val[FL_THREADS / 64][3 * FL_QCAP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * FL_TX + lane;
    const int ybase = (blockIdx.y * (FL_THREADS / 64) + wv) * FL_ROWS;
    StrayQueue q{q_addr[wv], q_val[wv], 0, (size_t)H * W};
    const int b = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const float v = v_loss[0];
    const float S1 = sums[1] + 1e-8f, S2 = sums[3] + 1e-8f;
    const float dN1 = v / S1, dN2 = v / S2;
    const float dS1 = -v * sums[0] / (S1 * S1), dS2 = -v * sums[2] / (S2 * S2);
    // d (coordinate after unnormalise) / d (pixel coordinate) = (size / 2) * 2 / (size - 1)
    const float kx = (0.5f * (float)W) * (2.f / (float)(W - 1)), ky = (0.5f * (float)H) * (2.f / (float)(H - 1));
    const float* ob = ori + (size_t)b * 3 * plane;
    float o[FL_ROWS][3], dav[FL_ROWS], go[FL_ROWS][3], gda[FL_ROWS];
#pragma unroll
    for (int r = 0; r < FL_ROWS; ++r) {
        const int y = ybase + r;
        const bool live = x < W && y < H;
        const size_t pix = live ? (size_t)y * W + x : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[r][c] = live ? ob[c * plane + pix] : 0.f;
            go[r][c] = 0.f;
        }
        dav[r] = live ? da[(size_t)b * plane + pix] : 0.f;
        gda[r] = 0.f;
    }
    for (int k = 0; k < K; ++k) {
        const size_t bk = (size_t)b * K + k;
        const float* lb = latent + bk * 3 * plane;
        Scatter<COMBINE> sc1, sc2;
        sc1.init();
        sc2.init();
#pragma unroll
        for (int r = 0; r < FL_ROWS; ++r) {
            const int y = ybase + r;
            const bool live = x < W && y < H;
            const size_t pix = live ? (size_t)y * W + x : 0;
            Taps t1 = Taps{0, 0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, false, false}, t2 = t1;
            float a1[3] = {0.f, 0.f, 0.f}, a2[3] = {0.f, 0.f, 0.f};
            bool act1 = false, act2 = false;
            if (live) {
                const float2 c1 = reinterpret_cast<const float2*>(e2m)[bk * plane + pix];
                const float2 c2 = reinterpret_cast<const float2*>(m2e)[bk * plane + pix];
                const float lav = la[bk * plane + pix];
                t1 = locate(c1.x, c1.y, W, H);
                t2 = locate(c2.x, c2.y, W, H);
                float gx1 = 0.f, gy1 = 0.f, gx2 = 0.f, gy2 = 0.f, gla = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const Quad q1 = fetch(ob + c * plane, t1, W);
                    const Quad q2 = fetch(lb + c * plane, t2, W);
                    const float d1 = bilinear(q1, t1) - lb[c * plane + pix];
                    const float d2 = bilinear(q2, t2) - o[r][c];
                    const float sg1 = sgn(d1 * lav), sg2 = sgn(d2 * dav[r]);
                    a1[c] = dN1 * sg1 * lav;
                    a2[c] = dN2 * sg2 * dav[r];
                    gla = __fmaf_rn(sg1, d1, gla);
                    gda[r] = __fmaf_rn(dN2 * sg2, d2, gda[r]);
                    go[r][c] -= a2[c];
                    coord_grad(q1, t1, a1[c], gx1, gy1);
                    coord_grad(q2, t2, a2[c], gx2, gy2);
                    if (d_latent) d_latent[bk * 3 * plane + c * plane + pix] = -a1[c];
                }
                gda[r] += 3.f * dS2;
                if (g_la) g_la[bk * plane + pix] = __fmaf_rn(dN1, gla, 3.f * dS1);
                if (g_e2m)
                    reinterpret_cast<float2*>(g_e2m)[bk * plane + pix] = make_float2(gx1 * t1.mx * kx, gy1 * t1.my * ky);
                if (g_m2e)
                    reinterpret_cast<float2*>(g_m2e)[bk * plane + pix] = make_float2(gx2 * t2.mx * kx, gy2 * t2.my * ky);
                act1 = a1[0] != 0.f || a1[1] != 0.f || a1[2] != 0.f;
                act2 = a2[0] != 0.f || a2[1] != 0.f || a2[2] != 0.f;
            }
            // (a pixel under a zero mask adds exactly zero: no atomics; its carried row, if any, leaves in row())
            if (g_ori) sc1.row(g_ori + (size_t)b * 3 * plane, plane, W, t1, act1, a1, lane, q);
            if (g_latent) sc2.row(g_latent + bk * 3 * plane, plane, W, t2, act2, a2, lane, q);
        }
        if (g_ori) sc1.flush(g_ori + (size_t)b * 3 * plane, plane, W, q, lane);
        if (g_latent) sc2.flush(g_latent + bk * 3 * plane, plane, W, q, lane);
    }
    q.drain(lane);
#pragma unroll
    for (int r = 0; r < FL_ROWS; ++r) {
        const int y = ybase + r;
        if (x < W && y < H) {
            const size_t pix = (size_t)y * W + x;
            if (g_da) g_da[(size_t)b * plane + pix] = gda[r];
            if (d_ori) {
#pragma unroll
                for (int c = 0; c < 3; ++c) d_ori[(size_t)b * 3 * plane + c * plane + pix] = go[r][c];
            }
        }
    }
}

// g += d over n floats (the direct image gradients join the scattered ones)
__global__ void __launch_bounds__(256) flow_add_kernel(size_t n4, float4* __restrict__ g, const float4* __restrict__ d,
                                                        size_t n, float* __restrict__ gs, const float* __restrict__ ds) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 a = g[i];
        const float4 b = d[i];
        a.x += b.x;
        a.y += b.y;
        a.z += b.z;
        a.w += b.w;
        g[i] = a;
    } else if (i == n4) {
        for (size_t j = 4 * n4; j < n; ++j) gs[j] += ds[j];
    }
}

}  // namespace mobgs

using namespace mobgs;

extern "C" {

int mobgs_flow_warp_loss_blocks(int B, int H, int W) {
    return B * ((H + FL_TY - 1) / FL_TY) * ((W + FL_TX - 1) / FL_TX);
}

static int flow_sizes_ok(const char* who, int B, int K, int H, int W) {
    if (B <= 0 || K <= 0 || H < 2 || W < 2) {
        set_error("%s: bad sizes B=%d K=%d H=%d W=%d (H, W >= 2: the coordinates are divided by size - 1)", who, B, K, H, W);
        return 0;
    }
    return 1;
}

int mobgs_flow_warp_loss_fwd(int B, int K, int H, int W, const float* ori, const float* latent, const float* exp2mid,
                             const float* mid2exp, const float* latent_alpha, const float* d_alpha, float* partial,
                             float* sums, float* loss, void* stream) {
    if (!flow_sizes_ok("mobgs_flow_warp_loss_fwd", B, K, H, W)) return MOBGS_E_INVALID;
    dim3 grid((W + FL_TX - 1) / FL_TX, (H + FL_TY - 1) / FL_TY, B);
    hipLaunchKernelGGL(flow_warp_l1_fwd_kernel, grid, dim3(FL_THREADS), 0, (hipStream_t)stream, B, K, H, W, ori, latent,
                       exp2mid, mid2exp, latent_alpha, d_alpha, partial);
    hipLaunchKernelGGL(flow_warp_l1_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       mobgs_flow_warp_loss_blocks(B, H, W), partial, sums, loss);
    return check_launch("flow_warp_l1_fwd_kernel");
}

size_t mobgs_flow_warp_loss_bwd_scratch_floats(int B, int K, int H, int W) {
    return (size_t)B * (K + 1) * 3 * (size_t)H * W;
}

int mobgs_flow_warp_loss_bwd(int B, int K, int H, int W, const float* ori, const float* latent, const float* exp2mid,
                             const float* mid2exp, const float* latent_alpha, const float* d_alpha, const float* sums,
                             const float* v_loss, float* g_ori, float* g_latent, float* g_exp2mid, float* g_mid2exp,
                             float* g_latent_alpha, float* g_d_alpha, float* scratch, int combine_taps, void* stream) {
    if (!flow_sizes_ok("mobgs_flow_warp_loss_bwd", B, K, H, W)) return MOBGS_E_INVALID;
    if ((g_ori || g_latent) && !scratch) {
        set_error("mobgs_flow_warp_loss_bwd: image gradients need the scratch buffer "
                  "(mobgs_flow_warp_loss_bwd_scratch_floats)");
        return MOBGS_E_INVALID;
    }
    const size_t plane = (size_t)H * W;
    float* d_ori = g_ori ? scratch : nullptr;                             // [B,3,H,W]
    float* d_latent = g_latent ? scratch + (size_t)B * 3 * plane : nullptr;  // [B,K,3,H,W]
    hipStream_t st = (hipStream_t)stream;
    auto launch = [&](auto kernel, int rows) {
        const int rows_per_wg = (FL_THREADS / 64) * rows;
        dim3 grid((W + FL_TX - 1) / FL_TX, (H + rows_per_wg - 1) / rows_per_wg, B);
        hipLaunchKernelGGL(kernel, grid, dim3(FL_THREADS), 0, st, B, K, H, W, ori, latent, exp2mid, mid2exp, latent_alpha,
                           d_alpha, sums, v_loss, g_ori, d_ori, g_latent, d_latent, g_exp2mid, g_mid2exp, g_latent_alpha,
                           g_d_alpha);
    };
    const bool tall = (size_t)((W + FL_TX - 1) / FL_TX) * ((H + 31) / 32) * B >= 1024;
    if (!combine_taps)
        launch(flow_warp_l1_bwd_kernel<false, 4>, 4);
    else if (tall)
        launch(flow_warp_l1_bwd_kernel<true, 8>, 8);
    else
        launch(flow_warp_l1_bwd_kernel<true, 4>, 4);
    auto add = [&](float* g, const float* d, size_t n) {
        // (both come from one allocation each: 16-byte aligned when the element offsets are multiples of 4)
        const bool aligned = ((uintptr_t)g % 16 == 0) && ((uintptr_t)d % 16 == 0);
        const size_t n4 = aligned ? n / 4 : 0;
        const size_t threads = n4 + 1;
        hipLaunchKernelGGL(flow_add_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, n4,
                           reinterpret_cast<float4*>(g), reinterpret_cast<const float4*>(d), n, g, d);
    };
    if (g_ori) add(g_ori, d_ori, (size_t)B * 3 * plane);
    if (g_latent) add(g_latent, d_latent, (size_t)B * K * 3 * plane);
    return check_launch("flow_warp_l1_bwd_kernel");
}

}  // extern "C"
