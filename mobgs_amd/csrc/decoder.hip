// Per-pixel colour decoder + expected-depth normalisation, forward and backward, for gfx950.
//
// Restates for the GPU:
//   /root/reference/helper_model.py:19-28                 Sandwich.forward: albedo|spec|timefeat = chunk(feat, 3);
//                                                         rgb = sigmoid(albedo + W2 relu(W1 [spec|timefeat|rays]))
//   gsplat rendering.py "ED" post-process [upstream]      depth = acc_depth / clamp(alpha, 1e-10)
//   /root/reference/gaussian_renderer/__init__.py:216-227 depth = img[..., -1]; feat = img[..., :-1].permute(0,3,1,2)
//
// Camera rays: the reference keeps a [1,6,H,W] map per camera (origin + unit view direction through each pixel
// centre, /root/reference/scene/cameras.py:132-146) -- 33 MB at 1352x1014, rebuilt for each of the 9 BLCE-warped
// cameras of a blurry view.  When the caller passes the pinhole parameters instead
// (`ray_intr` = {fx, fy, cx, cy}, `ray_c2w` = the first three rows of the row-major camera-to-world matrix, `rays` =
// NULL) the rays are generated in registers and the backward pass reduces their gradient straight into the 12
// entries of c2w.
//
// One thread per pixel; the 90 weights live in SGPRs (wave-uniform, scalar loads).  Reads the compositor's
// channels-last image [H,W,10] (one 40-byte row per lane), the planar ray map [6,H,W] and writes planar rgb
// [3,H,W] + depth [H,W]: every access is a unit-stride stream.  HBM-bound: 68 B read + 16 B written per pixel.
#include "common.h"
#include "decoder_shared.h"

namespace mobgs {

constexpr int DEC_THREADS = 256;
#ifndef MOBGS_DEC_W1_ROWS
#define MOBGS_DEC_W1_ROWS 2
#endif
constexpr int W1_ROWS = MOBGS_DEC_W1_ROWS;  // rows of the first layer held in SGPRs at a time (see reload_here)
constexpr int NRED = 102;  // per-workgroup partial row: 72 (w1) + 18 (w2) + 12 (c2w) gradients

struct Weights {
    float w1[72];  // [6][12]
    float w2[18];  // [3][6]
};

__device__ inline Weights load_weights(const float* __restrict__ w1, const float* __restrict__ w2) {
    Weights W;
#pragma unroll
    for (int k = 0; k < 72; ++k) W.w1[k] = w1[k];
#pragma unroll
    for (int k = 0; k < 18; ++k) W.w2[k] = w2[k];
    return W;
}

// per-image strides of a batched launch (grid.y = images; all 0 for one image): floats between consecutive images' ray
// maps / intrinsics / poses (0 = shared by all images)
struct DecBatch {
    size_t rays_stride;
    int intr_stride, c2w_stride;
    // channels [c0, c0 + cn) of the image that the decoder does not read, handed out as a tensor of their own [C,P,cn]
    // (forward: chan_out) / their cotangent taken in (backward: v_chan) -- get_flow()'s two flow channels ride in the
    // 12-channel exposure image; slicing them off in PyTorch is a strided copy of the whole image each way
    float* chan_out = nullptr;
    const float* v_chan = nullptr;
    int c0 = 0, cn = 0;
};

// feat_hw: [P, CF] channels-last with CF >= 9 (+1 accumulated depth when has_depth)
__global__ void __launch_bounds__(DEC_THREADS)
decoder_fwd_kernel(int P, int CF, int has_depth, int width, const float* __restrict__ feat_hw,
                   const float* __restrict__ alphas, const float* __restrict__ rays,
                   const float* __restrict__ ray_intr, const float* __restrict__ ray_c2w,
                   const float* __restrict__ w1, const float* __restrict__ w2,
                   float* __restrict__ rgb, float* __restrict__ depth, DecBatch bt) {
    {   // image blockIdx.y of a batch (one launch decodes the K sub-frames of a blurry view)
        const size_t cb = blockIdx.y;
        feat_hw += cb * P * CF;
        if (alphas) alphas += cb * P;
        if (rays) rays += cb * bt.rays_stride;
        rgb += cb * 3 * P;
        if (depth) depth += cb * P;
    }
    RayCam cam;
    if (!rays) cam = load_raycam(ray_intr + blockIdx.y * bt.intr_stride, ray_c2w + blockIdx.y * bt.c2w_stride);
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const float* f = feat_hw + (size_t)p * CF;
        float fr[10], r[6], out[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) fr[k] = f[k];
        fr[9] = 0.f;
        if (rays) {
#pragma unroll
            for (int k = 0; k < 6; ++k) r[k] = rays[(size_t)k * P + p];
        } else {
            float loc[2], inv_n;
            pixel_ray(cam, p, width, r, loc, inv_n);
        }
        sandwich_forward(w1, w2, fr, r, out);
#pragma unroll
        for (int o = 0; o < 3; ++o) rgb[(size_t)o * P + p] = out[o];
        if (has_depth) depth[p] = f[9] / fmaxf(alphas[p], 1e-10f);
        if (bt.chan_out)
            for (int k = 0; k < bt.cn; ++k) bt.chan_out[((size_t)blockIdx.y * P + p) * bt.cn + k] = f[bt.c0 + k];
    }
}

__device__ inline float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ void dec_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v_feat_hw [P, CF] is fully written (channels >= 10 get 0); v_alphas [P] written when has_depth;
// v_rays [6,P] written when non-null; weight gradients go to w_partial [gridDim.x, NRED] (summed by the next kernel).
//
// The first-layer weight gradient g_w1[j][c] = sum_pixels vh[j] * x[c] (6 x 12, 72 of the 90 weight gradients) is
// a [6 x P] x [P x 12] product: it runs on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, 4 pixels per
// instruction), each wave staging the vh / x rows of its 64 pixels in LDS so that lane l can supply
// A[l % 16][l / 16] and B[l / 16][l % 16].  That removes 72 per-thread accumulators (the kernel was limited to 2
// waves per SIMD by its registers) and their 72 end-of-kernel wave reductions; the small second layer (18) and the
// camera gradient (12) stay per-thread accumulators.  Measured 68 -> 60 us at 1352x1014.  (Also tried, both slower:
// staging the channels-last rows through LDS for 16-byte global accesses, 64 us; weights in LDS instead of SGPRs --
// the 90 weights + camera + pointers exceed the SGPR file and are partly spilled to VGPR lanes -- 87 us, because the
// register allocator then keeps the uniform values in VGPRs and the occupancy halves.)
constexpr int NACC = 30;  // 18 (w2) + 12 (c2w)
// RAY_MAP: the rays come from the [6,P] map (else from the pinhole parameters, in registers).  A template parameter, not
// a run-time branch: with both paths in one kernel the wait for the map's loads is placed on the joined path and, the
// vector-memory counter being in order, drains the prefetch of the next iteration with it.
template <bool RAY_MAP>
__global__ void __launch_bounds__(DEC_THREADS)
decoder_bwd_kernel(int P, int CF, int has_depth, int width, const float* __restrict__ feat_hw,
                   const float* __restrict__ alphas, const float* __restrict__ rays,
                   const float* __restrict__ ray_intr, const float* __restrict__ ray_c2w, int want_cam_grad,
                   const float* __restrict__ w1,
                   const float* __restrict__ w2, const float* __restrict__ v_rgb, const float* __restrict__ v_depth,
                   float* __restrict__ v_feat_hw, float* __restrict__ v_alphas, float* __restrict__ v_rays,
                   float* __restrict__ w_partial, DecBatch bt) {
    __shared__ __attribute__((aligned(16))) float s_a[DEC_THREADS / 64][64][8];    // vh[6] (+2 zeros) per pixel
    __shared__ __attribute__((aligned(16))) float s_b[DEC_THREADS / 64][64][12];   // x[12] per pixel
    {   // image blockIdx.y of a batch; its partial rows follow those of the images before it
        const size_t cb = blockIdx.y;
        feat_hw += cb * P * CF;
        if (alphas) alphas += cb * P;
        if (RAY_MAP) rays += cb * bt.rays_stride;
        v_rgb += cb * 3 * P;
        if (v_depth) v_depth += cb * P;
        v_feat_hw += cb * P * CF;
        if (v_alphas) v_alphas += cb * P;
        if (v_rays) v_rays += cb * 6 * P;
        w_partial += cb * gridDim.x * NRED;
    }
    RayCam cam;
    if (!RAY_MAP) cam = load_raycam(ray_intr + blockIdx.y * bt.intr_stride, ray_c2w + blockIdx.y * bt.c2w_stride);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;  // MFMA operand coordinates of this lane
    float gw[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) gw[k] = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};  // C[row = 4 * kq + i][col = m] = sum_px vh[row] * x[col]
    const int stride = gridDim.x * blockDim.x;
    const int p0 = blockIdx.x * blockDim.x + threadIdx.x;
    const int iters = (P - (blockIdx.x * blockDim.x + wv * 64) + stride - 1) / stride;  // wave-uniform trip count
    // Software pipeline: the kernel moves 104 B per pixel with 4 waves per SIMD and one dependent chain per iteration
    // (loads -> 300 instructions -> stores -> LDS -> 16 MFMAs), i.e. it is bound by bytes in flight, not by issue: the
    // per-pixel inputs of iteration it + 1 are requested before iteration it is evaluated.
    struct PixelIn {
        float f[10], v_rgb[3], alpha, v_depth;
    };
    // (every load unconditional, from clamped addresses: a branch around a load puts the register shuffle of the
    // merged value -- and with it the wait for the load -- right behind the load)
    const float* alpha_src = has_depth ? alphas : feat_hw;
    const float* vdepth_src = (has_depth && v_depth) ? v_depth : feat_hw;
    const int depth_ch = has_depth ? 9 : 0;
    auto fetch = [&](int p, PixelIn& in) {
        p = min(p, P - 1);
        const float* f = feat_hw + (size_t)p * CF;
#pragma unroll
        for (int k = 0; k < 9; ++k) in.f[k] = f[k];
        in.f[9] = f[depth_ch];
#pragma unroll
        for (int o = 0; o < 3; ++o) in.v_rgb[o] = v_rgb[(size_t)o * P + p];
        in.alpha = alpha_src[has_depth ? p : 0];
        in.v_depth = vdepth_src[(has_depth && v_depth) ? p : 0];
    };
    PixelIn next;
    fetch(p0, next);
    for (int it = 0; it < iters; ++it) {
        const int p = p0 + it * stride;
        const bool live = p < P;
        const PixelIn in = next;
        fetch(p + stride, next);
        float x[12], vh[6];
#pragma unroll
        for (int k = 0; k < 12; ++k) x[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) vh[k] = 0.f;
        if (live) {
            const float* f = in.f;
            float loc[2] = {0.f, 0.f}, inv_n = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) x[k] = f[3 + k];
            if constexpr (RAY_MAP) {
#pragma unroll
                for (int k = 0; k < 6; ++k) x[6 + k] = rays[(size_t)k * P + p];
            } else {
                pixel_ray(cam, p, width, x + 6, loc, inv_n);
            }
            float h[6];
#pragma unroll
            for (int jg = 0; jg < 6 / W1_ROWS; ++jg) {
                const ConstWeights w1a = reload_here(w1 + 12 * W1_ROWS * jg);
#pragma unroll
                for (int jj = 0; jj < W1_ROWS; ++jj) {
                    float s = 0.f;
#pragma unroll
                    for (int c = 0; c < 12; ++c) s = __fmaf_rn(w1a[12 * jj + c], x[c], s);
                    h[W1_ROWS * jg + jj] = fmaxf(s, 0.f);
                }
            }
            const ConstWeights w2a = reload_here(w2);
            float vy[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                float y = 0.f;
#pragma unroll
                for (int j = 0; j < 6; ++j) y = __fmaf_rn(w2a[6 * o + j], h[j], y);
                const float sg = 1.f / (1.f + __expf(-(f[o] + y)));
                vy[o] = in.v_rgb[o] * sg * (1.f - sg);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float s = 0.f;
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    s = __fmaf_rn(w2a[6 * o + j], vy[o], s);
                    gw[6 * o + j] = __fmaf_rn(vy[o], h[j], gw[6 * o + j]);
                }
                vh[j] = h[j] > 0.f ? s : 0.f;
            }
            float vx[12];
#pragma unroll
            for (int c = 0; c < 12; ++c) vx[c] = 0.f;
#pragma unroll
            for (int jg = 0; jg < 6 / W1_ROWS; ++jg) {  // (j ascending per component, as one chain of FMAs)
                const ConstWeights w1b = reload_here(w1 + 12 * W1_ROWS * jg);
#pragma unroll
                for (int jj = 0; jj < W1_ROWS; ++jj)
#pragma unroll
                    for (int c = 0; c < 12; ++c) vx[c] = __fmaf_rn(w1b[12 * jj + c], vh[W1_ROWS * jg + jj], vx[c]);
            }
            float* vf = v_feat_hw + (size_t)p * CF;
#pragma unroll
            for (int o = 0; o < 3; ++o) vf[o] = vy[o];
#pragma unroll
            for (int k = 0; k < 6; ++k) vf[3 + k] = vx[k];
            if (has_depth) {
                const float a = in.alpha;
                const float ac = fmaxf(a, 1e-10f);
                const float g = v_depth ? in.v_depth : 0.f;
                vf[9] = g / ac;
                v_alphas[p] = a > 1e-10f ? -g * f[9] / (ac * ac) : 0.f;
            }
            for (int k = 9 + (has_depth ? 1 : 0); k < CF; ++k)
                vf[k] = (bt.v_chan && k >= bt.c0 && k < bt.c0 + bt.cn)
                            ? bt.v_chan[((size_t)blockIdx.y * P + p) * bt.cn + (k - bt.c0)] : 0.f;
            if (v_rays) {
#pragma unroll
                for (int k = 0; k < 6; ++k) v_rays[(size_t)k * P + p] = vx[6 + k];
            }
            if (!RAY_MAP && want_cam_grad) {
                // origin = t; dir = d / |d| with d = R loc + [third column]: v_d = (v_dir - dir <dir, v_dir>) / |d|
                const float dotp = x[9] * vx[9] + x[10] * vx[10] + x[11] * vx[11];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float vd = (vx[9 + i] - x[9 + i] * dotp) * inv_n;
                    gw[18 + 4 * i] += vd * loc[0];
                    gw[18 + 4 * i + 1] += vd * loc[1];
                    gw[18 + 4 * i + 2] += vd;
                    gw[18 + 4 * i + 3] += vx[6 + i];
                }
            }
        }
        // g_w1 += vh^T x over this wave's 64 pixels (pixels past the end contribute vh = 0)
        dec_wave_fence();  // the previous iteration's operand reads are done
        reinterpret_cast<float4*>(s_a[wv][lane])[0] = make_float4(vh[0], vh[1], vh[2], vh[3]);
        reinterpret_cast<float4*>(s_a[wv][lane])[1] = make_float4(vh[4], vh[5], 0.f, 0.f);
        reinterpret_cast<float4*>(s_b[wv][lane])[0] = make_float4(x[0], x[1], x[2], x[3]);
        reinterpret_cast<float4*>(s_b[wv][lane])[1] = make_float4(x[4], x[5], x[6], x[7]);
        reinterpret_cast<float4*>(s_b[wv][lane])[2] = make_float4(x[8], x[9], x[10], x[11]);
        dec_wave_fence();
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const int px = 4 * s4 + kq;
            const float av = m < 8 ? s_a[wv][px][m & 7] : 0.f;
            const float bv = m < 12 ? s_b[wv][px][m < 12 ? m : 0] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
    }
    // per-workgroup partial row: [0,72) g_w1 from the MFMA accumulators of the 4 waves, [72,90) g_w2, [90,102) c2w
    __shared__ float red[DEC_THREADS / 64][NRED];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * kq + i;
        if (row < 6 && m < 12) red[wv][12 * row + m] = acc[i];
    }
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        const float sum = wave_sum_f(gw[k]);
        if (lane == 0) red[wv][72 + k] = sum;
    }
    __syncthreads();
    if (threadIdx.x < NRED) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < DEC_THREADS / 64; ++w) sum += red[w][threadIdx.x];
        w_partial[(size_t)blockIdx.x * NRED + threadIdx.x] = sum;
    }
}

// one workgroup per weight component: 102 workgroups x 256 threads sum the per-workgroup partial rows in a fixed
// order.  `accumulate`: the weight gradients are ADDED to g_w1 / g_w2 (several renders of one backward pass writing
// into the same .grad).  Workgroups past the 102nd clear the fourth row of a 4 x 4 pose gradient.
// A batch of C images (grid.y): the weight gradients are sums over the rows of ALL images (workgroups with
// blockIdx.y = 0), the pose gradient of image c over that image's rows only (g_c2w + c * c2w_floats).
__global__ void __launch_bounds__(256) decoder_wgrad_reduce_kernel(int nblocks, const float* __restrict__ w_partial,
                                                                     float* __restrict__ g_w1,
                                                                     float* __restrict__ g_w2,
                                                                     float* __restrict__ g_c2w, int accumulate,
                                                                     int c2w_floats) {
    const int k = blockIdx.x;
    const int img = blockIdx.y;
    if (g_c2w) g_c2w += (size_t)img * c2w_floats;
    if (k >= NRED) {
        if (threadIdx.x == 0) g_c2w[k - 90] = 0.f;
        return;
    }
    int first = 0, count = nblocks * (int)gridDim.y;     // weights: every row
    if (k >= 90) {                                        // pose: the rows of this image
        first = img * nblocks;
        count = nblocks;
    } else if (img != 0) {
        return;
    }
    w_partial += (size_t)first * NRED;
    float s = 0.f;
    for (int b = threadIdx.x; b < count; b += 256) s += w_partial[(size_t)b * NRED + k];
    s = wave_sum_f(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = red[0] + red[1] + red[2] + red[3];
        if (k < 72)
            g_w1[k] = accumulate ? g_w1[k] + t : t;
        else if (k < 90)
            g_w2[k - 72] = accumulate ? g_w2[k - 72] + t : t;
        else if (g_c2w)
            g_c2w[k - 90] = t;
    }
}

// The same sums for the MANY rows of the backward compositor's decoder prologue (raster.hip DECB: one row per tile, 5440 per
// image at 1352x1014; the kernel above walks one column per workgroup -- every workgroup touches every cache line of the
// rows: 10 us there).  Here workgroup (b, img) streams a contiguous chunk of image img's rows (thread = column, two row
// phases: coalesced 408-byte rows) into chunk_sums[img * G + b][102]; the last workgroup to finish (a ticket) adds the chunk
// sums in a fixed order: deterministic, one launch.  The ticket word is zeroed by the compositing launch in front of it.
constexpr int WRED_CHUNKS_MAX = 64;
__host__ __device__ inline int wred_chunks(int rows_per_image) { return min(WRED_CHUNKS_MAX, (rows_per_image + 31) / 32); }

constexpr int WRED_THREADS = 1024, WRED_PHASES = WRED_THREADS / 128;
__global__ void __launch_bounds__(WRED_THREADS) decoder_wgrad_reduce_rows_kernel(int rows_per_image, const float* __restrict__ w_partial,
                                                                          float* __restrict__ chunk_sums,
                                                                          unsigned* __restrict__ ticket,
                                                                          float* __restrict__ g_w1, float* __restrict__ g_w2,
                                                                          float* __restrict__ g_c2w, int accumulate,
                                                                          int c2w_floats) {
    const int G = gridDim.x, C = gridDim.y;
    const int b = blockIdx.x, img = blockIdx.y;
    const int per = (rows_per_image + G - 1) / G;
    const int r0 = img * rows_per_image + b * per;
    const int r1 = min(r0 + per, (img + 1) * rows_per_image);
    const int col = threadIdx.x & 127, ph = threadIdx.x >> 7;
    float s = 0.f;
    if (col < NRED) {
#pragma unroll 4
        for (int r = r0 + ph; r < r1; r += WRED_PHASES) s += w_partial[(size_t)r * NRED + col];
    }
    __shared__ float red[WRED_PHASES][128];
    __shared__ int is_last;
    red[ph][col] = s;
    __syncthreads();
    if (threadIdx.x < NRED) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < WRED_PHASES; ++q) t += red[q][threadIdx.x];
        // relaxed agent-scope atomics for the words that cross workgroups (they go to the coherence point, past the XCD's own
        // L2): a release / acquire fence pair instead writes back and invalidates that whole L2 per workgroup -- 25 us for
        // the 64 workgroups of one image (isect.hip scan_lookback found the same)
        __hip_atomic_store(&chunk_sums[(size_t)(img * G + b) * NRED + threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0);   // the stores have left before the workgroup takes its ticket (the barrier orders the waves)
    __syncthreads();
    if (threadIdx.x == 0)
        is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(G * C - 1);
    __syncthreads();
    if (!is_last) return;
    // image by image: the G chunk sums of an image staged in LDS by all threads at once (one dependent load each -- a thread
    // summing its column straight from memory pays a memory round trip per chunk: 33 us), then added in chunk order
    __shared__ float stage[WRED_CHUNKS_MAX * NRED];
    const int k = threadIdx.x;
    float wsum = 0.f;
    for (int im = 0; im < C; ++im) {
        __syncthreads();
        for (int i = threadIdx.x; i < G * NRED; i += WRED_THREADS)
            stage[i] = __hip_atomic_load(&chunk_sums[(size_t)im * G * NRED + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (k < NRED) {
            float t = 0.f;
            for (int i = 0; i < G; ++i) t += stage[i * NRED + k];
            if (k < 90)
                wsum += t;
            else if (g_c2w)
                g_c2w[(size_t)im * c2w_floats + (k - 90)] = t;
        } else if (g_c2w && c2w_floats == 16 && k < NRED + 4) {
            g_c2w[(size_t)im * c2w_floats + (k - 90)] = 0.f;
        }
    }
    if (k < 72)
        g_w1[k] = accumulate ? g_w1[k] + wsum : wsum;
    else if (k < 90)
        g_w2[k - 72] = accumulate ? g_w2[k - 72] + wsum : wsum;
    if (k == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// decoder_shared.h: the fixed-order sum of the prologue's partial rows; `scratch` = decoder_wgrad_scratch_floats(C, rows) floats
size_t decoder_wgrad_scratch_floats(int C, int rows_per_image) {
    return (size_t)C * rows_per_image * NRED + (size_t)C * wred_chunks(rows_per_image) * NRED + 4;
}
unsigned* decoder_wgrad_ticket(float* scratch, int C, int rows_per_image) {
    return reinterpret_cast<unsigned*>(scratch + decoder_wgrad_scratch_floats(C, rows_per_image) - 4);
}
void launch_decoder_wgrad_reduce(int C, int rows_per_image, float* scratch, float* g_w1, float* g_w2, float* g_c2w,
                                 int g_c2w_floats, int accumulate, hipStream_t st) {
    float* chunk_sums = scratch + (size_t)C * rows_per_image * NRED;
    hipLaunchKernelGGL(decoder_wgrad_reduce_rows_kernel, dim3(wred_chunks(rows_per_image), C), dim3(WRED_THREADS), 0, st,
                       rows_per_image, scratch, chunk_sums, decoder_wgrad_ticket(scratch, C, rows_per_image), g_w1, g_w2,
                       g_c2w, accumulate, g_c2w_floats);
}

}  // namespace mobgs

using namespace mobgs;

extern "C" {

static int decoder_grid(int P) {
    int g = (P + DEC_THREADS * 3 - 1) / (DEC_THREADS * 3);  // ~3 pixels per thread (measured optimum, 3..12 within 4 %)
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    return g;
}

int mobgs_decoder_bwd_blocks(int P) { return decoder_grid(P); }

static int decoder_args_ok(const char* who, int C, int P, int CF, int has_depth, int width, const float* rays,
                           const float* ray_intr, const float* ray_c2w) {
    if (C < 1 || C > 65535 || P < 0 || CF < 9 + (has_depth ? 1 : 0) || (!rays && (!ray_intr || !ray_c2w || width <= 0))) {
        set_error("%s: bad arguments C=%d P=%d CF=%d (rays or ray_intr+ray_c2w+width required)", who, C, P, CF);
        return 0;
    }
    return 1;
}

static bool channel_args_ok(const char* who, int CF, int has_depth, const void* chan, int c0, int n) {
    if (!chan) return true;
    if (n < 1 || c0 < 9 + (has_depth ? 1 : 0) || c0 + n > CF) {
        set_error("%s: extra channels [%d, %d) must lie behind the channels the decoder reads and inside CF = %d", who, c0,
                  c0 + n, CF);
        return false;
    }
    return true;
}

int mobgs_decoder_fwd_channels(int C, int P, int CF, int has_depth, int width, const float* feat_hw, const float* alphas,
                               const float* rays, int64_t rays_stride, const float* ray_intr, int intr_stride,
                               const float* ray_c2w, int c2w_stride, const float* w1, const float* w2, float* rgb,
                               float* depth, float* chan_out, int c0, int n, void* stream) {
    if (!decoder_args_ok("mobgs_decoder_fwd", C, P, CF, has_depth, width, rays, ray_intr, ray_c2w) ||
        !channel_args_ok("mobgs_decoder_fwd_channels", CF, has_depth, chan_out, c0, n))
        return MOBGS_E_INVALID;
    if (P == 0) return MOBGS_OK;
    int g = (P + DEC_THREADS - 1) / DEC_THREADS;
    if (g > 4096) g = 4096;
    DecBatch bt{(size_t)rays_stride, intr_stride, c2w_stride};
    if (chan_out) {
        bt.chan_out = chan_out;
        bt.c0 = c0;
        bt.cn = n;
    }
    hipLaunchKernelGGL(decoder_fwd_kernel, dim3(g, C), dim3(DEC_THREADS), 0, (hipStream_t)stream, P, CF, has_depth, width,
                       feat_hw, alphas, rays, ray_intr, ray_c2w, w1, w2, rgb, depth, bt);
    return check_launch("decoder_fwd_kernel");
}

int mobgs_decoder_fwd_many(int C, int P, int CF, int has_depth, int width, const float* feat_hw, const float* alphas,
                           const float* rays, int64_t rays_stride, const float* ray_intr, int intr_stride,
                           const float* ray_c2w, int c2w_stride, const float* w1, const float* w2, float* rgb,
                           float* depth, void* stream) {
    return mobgs_decoder_fwd_channels(C, P, CF, has_depth, width, feat_hw, alphas, rays, rays_stride, ray_intr, intr_stride,
                                      ray_c2w, c2w_stride, w1, w2, rgb, depth, nullptr, 0, 0, stream);
}

int mobgs_decoder_fwd(int P, int CF, int has_depth, int width, const float* feat_hw, const float* alphas,
                      const float* rays, const float* ray_intr, const float* ray_c2w, const float* w1,
                      const float* w2, float* rgb, float* depth, void* stream) {
    return mobgs_decoder_fwd_many(1, P, CF, has_depth, width, feat_hw, alphas, rays, 0, ray_intr, 0, ray_c2w, 0, w1, w2,
                                  rgb, depth, stream);
}

int mobgs_decoder_bwd_many(int C, int P, int CF, int has_depth, int width, const float* feat_hw, const float* alphas,
                           const float* rays, int64_t rays_stride, const float* ray_intr, int intr_stride,
                           const float* ray_c2w, int c2w_stride, const float* w1, const float* w2, const float* v_rgb,
                           const float* v_depth, float* v_feat_hw, float* v_alphas, float* v_rays, float* w_partial,
                           float* g_w1, float* g_w2, float* g_c2w, int g_c2w_floats, int accumulate_wgrad,
                           void* stream) {
    return mobgs_decoder_bwd_channels(C, P, CF, has_depth, width, feat_hw, alphas, rays, rays_stride, ray_intr, intr_stride,
                                      ray_c2w, c2w_stride, w1, w2, v_rgb, v_depth, v_feat_hw, v_alphas, v_rays, w_partial,
                                      g_w1, g_w2, g_c2w, g_c2w_floats, accumulate_wgrad, nullptr, 0, 0, stream);
}

int mobgs_decoder_bwd_channels(int C, int P, int CF, int has_depth, int width, const float* feat_hw, const float* alphas,
                               const float* rays, int64_t rays_stride, const float* ray_intr, int intr_stride,
                               const float* ray_c2w, int c2w_stride, const float* w1, const float* w2, const float* v_rgb,
                               const float* v_depth, float* v_feat_hw, float* v_alphas, float* v_rays, float* w_partial,
                               float* g_w1, float* g_w2, float* g_c2w, int g_c2w_floats, int accumulate_wgrad,
                               const float* v_chan, int c0, int n, void* stream) {
    if (!channel_args_ok("mobgs_decoder_bwd_channels", CF, has_depth, v_chan, c0, n)) return MOBGS_E_INVALID;
    if (!decoder_args_ok("mobgs_decoder_bwd", C, P, CF, has_depth, width, rays, ray_intr, ray_c2w) || P == 0 ||
        (g_c2w && g_c2w_floats != 12 && g_c2w_floats != 16)) {
        if (P == 0) set_error("mobgs_decoder_bwd: P = 0");
        return MOBGS_E_INVALID;
    }
    if (C > 1 && ((v_rays && rays_stride == 0) || (g_c2w && c2w_stride == 0))) {
        set_error("mobgs_decoder_bwd_many: a ray map / pose shared by the images of a batch cannot receive a gradient "
                  "(one gradient per image is written)");
        return MOBGS_E_INVALID;
    }
    const int g = decoder_grid(P);
    DecBatch bt{(size_t)rays_stride, intr_stride, c2w_stride};
    if (v_chan) {
        bt.v_chan = v_chan;
        bt.c0 = c0;
        bt.cn = n;
    }
    hipLaunchKernelGGL(rays ? decoder_bwd_kernel<true> : decoder_bwd_kernel<false>, dim3(g, C), dim3(DEC_THREADS), 0,
                       (hipStream_t)stream, P, CF, has_depth, width,
                       feat_hw, alphas, rays, ray_intr, ray_c2w, g_c2w ? 1 : 0, w1, w2, v_rgb, v_depth, v_feat_hw,
                       v_alphas, v_rays, w_partial, bt);
    const int nred = NRED + ((g_c2w && g_c2w_floats == 16) ? 4 : 0);
    hipLaunchKernelGGL(decoder_wgrad_reduce_kernel, dim3(nred, C), dim3(256), 0, (hipStream_t)stream, g, w_partial, g_w1,
                       g_w2, g_c2w, accumulate_wgrad, g_c2w_floats);
    return check_launch("decoder_bwd_kernel");
}

int mobgs_decoder_bwd(int P, int CF, int has_depth, int width, const float* feat_hw, const float* alphas,
                      const float* rays, const float* ray_intr, const float* ray_c2w, const float* w1,
                      const float* w2, const float* v_rgb, const float* v_depth, float* v_feat_hw, float* v_alphas,
                      float* v_rays, float* w_partial, float* g_w1, float* g_w2, float* g_c2w, int g_c2w_floats,
                      int accumulate_wgrad, void* stream) {
    return mobgs_decoder_bwd_many(1, P, CF, has_depth, width, feat_hw, alphas, rays, 0, ray_intr, 0, ray_c2w, 0, w1, w2,
                                  v_rgb, v_depth, v_feat_hw, v_alphas, v_rays, w_partial, g_w1, g_w2, g_c2w,
                                  g_c2w_floats, accumulate_wgrad, stream);
}

}  // extern "C"
