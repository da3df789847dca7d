// Pieces of the Sandwich decoder shared by decoder.hip (its own kernels) and raster.hip (the decoder as the epilogue of
// the forward compositor): weight reloads through the scalar unit, the pinhole ray of a pixel, and ONE evaluation of
//   rgb = sigmoid(albedo + W2 relu(W1 [spec | timefeat | rays])),  depth = accumulated depth / max(alpha, 1e-10)
// (/root/reference/helper_model.py:19-28, gsplat's "ED" post-process) -- the same instruction sequence wherever it runs,
// so a fused and a separate decode give bit-identical images.
#pragma once
#include "common.h"

namespace mobgs {

constexpr int W1_ROWS_SHARED = 2;  // rows of the first layer held in SGPRs at a time (see reload_here)

// The 90 weights + the camera + the kernel's pointers do not fit the SGPR file: kept live across the pixel loop, the
// register allocator parks the excess in VGPR lanes and every iteration pays ~190 v_readlane_b32 (half-rate VALU) to get
// them back -- as many issue slots as the arithmetic.  Instead each use re-reads its weights with scalar loads from a
// pointer the optimiser cannot see through (so the loads stay inside the loop, next to their use): they hit the scalar
// cache and issue on the scalar unit, off the VALU.
typedef const float __attribute__((address_space(4))) * ConstWeights;  // constant address space: scalar loads
__device__ __forceinline__ ConstWeights reload_here(const float* p) {
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (ConstWeights)(((unsigned long long)hi << 32) | lo);
}

struct RayCam {
    float fx, fy, cx, cy;
    float c2w[12];  // row-major 3x4: [R | t], camera -> world
};
__device__ inline RayCam load_raycam(const float* __restrict__ intr, const float* __restrict__ c2w) {
    RayCam c;
    c.fx = intr[0]; c.fy = intr[1]; c.cx = intr[2]; c.cy = intr[3];
#pragma unroll
    for (int k = 0; k < 12; ++k) c.c2w[k] = c2w[k];
    return c;
}
// origin + normalised direction of pixel (px, py); also returns the local direction and 1/|d|.  Every product-sum is
// an explicit FMA: this runs in decoder.hip AND as the compositor's epilogue in raster.hip, two translation units with
// different vectoriser / contraction settings, and both must produce the same bits.
__device__ inline void pixel_ray_xy(const RayCam& c, int px, int py, float r[6], float loc[2], float& inv_n) {
    loc[0] = ((float)px + 0.5f - c.cx) / c.fx;
    loc[1] = ((float)py + 0.5f - c.cy) / c.fy;
    float d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = __fmaf_rn(c.c2w[4 * i], loc[0], __fmaf_rn(c.c2w[4 * i + 1], loc[1], c.c2w[4 * i + 2]));
    inv_n = 1.f / sqrtf(__fmaf_rn(d[0], d[0], __fmaf_rn(d[1], d[1], d[2] * d[2])));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        r[i] = c.c2w[4 * i + 3];
        r[3 + i] = d[i] * inv_n;
    }
}
__device__ inline void pixel_ray(const RayCam& c, int p, int W, float r[6], float loc[2], float& inv_n) {
    const int py = p / W, px = p - py * W;
    pixel_ray_xy(c, px, py, r, loc, inv_n);
}

// One pixel: f[0..9] = the composited features (+ accumulated depth in f[9] when has_depth), rays r[6] -> rgb[3]
__device__ __forceinline__ void sandwich_forward(const float* __restrict__ w1, const float* __restrict__ w2,
                                                 const float (&f)[10], const float (&r)[6], float (&rgb)[3]) {
    float x[12];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        x[k] = f[3 + k];
        x[6 + k] = r[k];
    }
    float h[6];
#pragma unroll
    for (int jg = 0; jg < 6 / W1_ROWS_SHARED; ++jg) {  // W1_ROWS_SHARED rows of W1 in SGPRs at a time
        const ConstWeights w1a = reload_here(w1 + 12 * W1_ROWS_SHARED * jg);
#pragma unroll
        for (int jj = 0; jj < W1_ROWS_SHARED; ++jj) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 12; ++c) s = __fmaf_rn(w1a[12 * jj + c], x[c], s);
            h[W1_ROWS_SHARED * jg + jj] = fmaxf(s, 0.f);
        }
    }
    const ConstWeights w2a = reload_here(w2);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) y = __fmaf_rn(w2a[6 * o + j], h[j], y);
        const float z = f[o] + y;
        rgb[o] = 1.f / (1.f + __expf(-z));
    }
}

// NPX pixels at once, WEIGHTS OUTER: each group of W1 rows (and W2) is fetched through the scalar unit once for all the
// pixels of the lane instead of once per pixel -- in the compositor's epilogue the wave has nothing else to hide those
// scalar-load round trips behind.  Per pixel the same FMA chains as sandwich_forward: bit-identical.
template <int NPX>
__device__ __forceinline__ void sandwich_forward_n(const float* __restrict__ w1, const float* __restrict__ w2,
                                                   const float (&f)[NPX][10], const float (&r)[NPX][6],
                                                   float (&rgb)[NPX][3]) {
    float h[NPX][6];
#pragma unroll
    for (int jg = 0; jg < 6 / W1_ROWS_SHARED; ++jg) {
        const ConstWeights w1a = reload_here(w1 + 12 * W1_ROWS_SHARED * jg);
#pragma unroll
        for (int jj = 0; jj < W1_ROWS_SHARED; ++jj) {
#pragma unroll
            for (int k = 0; k < NPX; ++k) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 6; ++c) s = __fmaf_rn(w1a[12 * jj + c], f[k][3 + c], s);
#pragma unroll
                for (int c = 0; c < 6; ++c) s = __fmaf_rn(w1a[12 * jj + 6 + c], r[k][c], s);
                h[k][W1_ROWS_SHARED * jg + jj] = fmaxf(s, 0.f);
            }
        }
    }
    const ConstWeights w2a = reload_here(w2);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
#pragma unroll
        for (int k = 0; k < NPX; ++k) {
            float y = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) y = __fmaf_rn(w2a[6 * o + j], h[k][j], y);
            const float z = f[k][o] + y;
            rgb[k][o] = 1.f / (1.f + __expf(-z));
        }
    }
}

// One pixel of the decoder's BACKWARD pass (decoder_bwd_kernel's arithmetic, instruction for instruction: explicit FMAs, the
// same chains in the same order, so that the backward compositor's prologue -- raster.hip, DECB -- hands the compositing
// loop bit for bit the cotangents a separate decoder_bwd launch would have written to memory):
//   in : f[0..9] composited features (+ accumulated depth in f[9]), x6[0..5] = ray origin | normalised direction,
//        v_rgb[3], alpha, g = cotangent of the expected depth (0 when unused)
//   out: vf[10] = cotangent of the composited image, v_alpha (from the depth normalisation), h[6] hidden activations,
//        vh[6] their cotangents, vy[3] pre-sigmoid cotangents, vdir[3] = cotangent of the normalised direction
__device__ __forceinline__ void sandwich_backward(const float* __restrict__ w1, const float* __restrict__ w2,
                                                  const float (&f)[10], const float (&x6)[6], const float (&v_rgb)[3],
                                                  float alpha, float g, float (&vf)[10], float& v_alpha, float (&h)[6],
                                                  float (&vh)[6], float (&vy)[3], float (&vdir)[3], float (&vorg)[3]) {
    float x[12];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        x[k] = f[3 + k];
        x[6 + k] = x6[k];
    }
#pragma unroll
    for (int jg = 0; jg < 6 / W1_ROWS_SHARED; ++jg) {
        const ConstWeights w1a = reload_here(w1 + 12 * W1_ROWS_SHARED * jg);
#pragma unroll
        for (int jj = 0; jj < W1_ROWS_SHARED; ++jj) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 12; ++c) s = __fmaf_rn(w1a[12 * jj + c], x[c], s);
            h[W1_ROWS_SHARED * jg + jj] = fmaxf(s, 0.f);
        }
    }
    const ConstWeights w2a = reload_here(w2);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) y = __fmaf_rn(w2a[6 * o + j], h[j], y);
        const float sg = 1.f / (1.f + __expf(-(f[o] + y)));
        vy[o] = v_rgb[o] * sg * (1.f - sg);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float s = 0.f;
#pragma unroll
        for (int o = 0; o < 3; ++o) s = __fmaf_rn(w2a[6 * o + j], vy[o], s);
        vh[j] = h[j] > 0.f ? s : 0.f;
    }
    float vx[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) vx[c] = 0.f;
#pragma unroll
    for (int jg = 0; jg < 6 / W1_ROWS_SHARED; ++jg) {  // (j ascending per component, as one chain of FMAs)
        const ConstWeights w1b = reload_here(w1 + 12 * W1_ROWS_SHARED * jg);
#pragma unroll
        for (int jj = 0; jj < W1_ROWS_SHARED; ++jj)
#pragma unroll
            for (int c = 0; c < 12; ++c) vx[c] = __fmaf_rn(w1b[12 * jj + c], vh[W1_ROWS_SHARED * jg + jj], vx[c]);
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) vf[o] = vy[o];
#pragma unroll
    for (int k = 0; k < 6; ++k) vf[3 + k] = vx[k];
    const float ac = fmaxf(alpha, 1e-10f);
    vf[9] = g / ac;
    v_alpha = alpha > 1e-10f ? -g * f[9] / (ac * ac) : 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        vorg[i] = vx[6 + i];
        vdir[i] = vx[9 + i];
    }
}

// One output element of the weight / pose gradient, by ONE workgroup of 256 threads: the fixed-order sum of column k of the
// partial rows [n_images * rows_per_image, 102] (rows of decoder_bwd_kernel or of the backward compositor's prologue) --
// k < 72: g_w1, < 90: g_w2 (over the rows of ALL images, by the workgroups with img = 0), < 102: the pose of image img (its
// own rows), 102..105: the fourth row of a 4 x 4 pose gradient (zeros).  Slow per workgroup (a strided column), which is why
// it runs BESIDE something else: as extra leading workgroups of the gradient-slot reduction (raster.hip, slot_reduce16).
struct WgradFinish {
    const float* w_partial = nullptr;
    float *g_w1 = nullptr, *g_w2 = nullptr, *g_c2w = nullptr;
    int rows_per_image = 0, n_images = 0, accumulate = 0, c2w_floats = 0;
};
constexpr int WGRAD_NRED = 102;
__host__ __device__ inline int wgrad_finish_outputs(const WgradFinish& f) {
    return WGRAD_NRED + ((f.g_c2w && f.c2w_floats == 16) ? 4 : 0);
}
__device__ inline void wgrad_column_sum(const WgradFinish& f, int k, int img) {
    float* g_c2w = f.g_c2w ? f.g_c2w + (size_t)img * f.c2w_floats : nullptr;
    if (k >= WGRAD_NRED) {
        if (threadIdx.x == 0) g_c2w[k - 90] = 0.f;
        return;
    }
    int first = 0, count = f.rows_per_image * f.n_images;   // weights: every row
    if (k >= 90) {                                            // pose: the rows of this image
        first = img * f.rows_per_image;
        count = f.rows_per_image;
    } else if (img != 0) {
        return;
    }
    const float* wp = f.w_partial + (size_t)first * WGRAD_NRED;
    float s = 0.f;
    for (int b = threadIdx.x; b < count; b += 256) s += wp[(size_t)b * WGRAD_NRED + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    __shared__ float wgrad_red[4];
    if ((threadIdx.x & 63) == 0) wgrad_red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = wgrad_red[0] + wgrad_red[1] + wgrad_red[2] + wgrad_red[3];
        if (k < 72)
            f.g_w1[k] = f.accumulate ? f.g_w1[k] + t : t;
        else if (k < 90)
            f.g_w2[k - 72] = f.accumulate ? f.g_w2[k - 72] + t : t;
        else if (g_c2w)
            g_c2w[k - 90] = t;
    }
}

// decoder.hip: sums the partial rows the backward compositor's decoder prologue left at the start of `scratch`
// ([C * rows_per_image, 102], decoder_bwd_kernel's row layout) in a fixed order into g_w1 [6,12], g_w2 [3,6] (over all
// images; accumulate != 0: added to what is there) and g_c2w [C, g_c2w_floats] (per image; may be NULL).  `scratch` holds
// decoder_wgrad_scratch_floats(C, rows_per_image) floats: the rows, the chunk sums, and a ticket word that must be ZERO
// when the reduction starts (the compositing kernel clears it) and is zero again when it ends.
size_t decoder_wgrad_scratch_floats(int C, int rows_per_image);
unsigned* decoder_wgrad_ticket(float* scratch, int C, int rows_per_image);
void launch_decoder_wgrad_reduce(int C, int rows_per_image, float* scratch, float* g_w1, float* g_w2, float* g_c2w,
                                 int g_c2w_floats, int accumulate, hipStream_t st);

}  // namespace mobgs
