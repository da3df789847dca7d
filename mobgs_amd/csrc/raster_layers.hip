// Layered compositing: the combined / static-only / dynamic-only renders of a train-mode render() in ONE pass.
//
// The reference's render(get_static=True, get_dynamic=True) rasterizes three splat sets with the same camera:
//   all = static | dynamic   /root/reference/gaussian_renderer/__init__.py:201-214
//   dynamic only             /root/reference/gaussian_renderer/__init__.py:143-156 (+ a ones-colour alpha pass :163-176)
//   static only              /root/reference/gaussian_renderer/__init__.py:236-249 (+ a ones-colour alpha pass :255-268)
// i.e. 5 gsplat rasterizations = 5 projections, 5 binning/sort passes, 5 compositing passes.  The per-tile list of
// the static (dynamic) subset is exactly the sub-sequence of the combined list with flat id < Ns (>= Ns), the
// per-splat projection does not depend on the set, and the two alpha passes equal (1 - T_final) + T_final * bg of
// the corresponding feature pass.  So one projection, one binning/sort and ONE walk over the combined list
// produce all three images: every splat is evaluated once per pixel and blended into layer 0 (all) and into the
// layer of its class; each layer keeps its own transmittance, stop flag and last index, which reproduces the
// three independent passes bit for bit.
//
// Mapping: 2 pixels per lane, two wave64 per 16x16 tile (rows 0-7 / 8-15): three layers of accumulators fit the
// register file at 4 waves per SIMD; backward writes one gradient record per (half tile, splat).
#include <type_traits>

#include "common.h"

namespace mobgs {

constexpr float L_ALPHA_MIN = 1.f / 255.f;
constexpr float L_ALPHA_MAX = 0.999f;
constexpr float L_T_STOP = 1e-4f;
constexpr int LPPL = 2;      // pixels per lane
constexpr int LWAVES = 4;    // waves per workgroup = 2 tiles x 2 halves
constexpr int NL = 3;        // layers: 0 = all, 1 = static (flat id % N < Ns), 2 = dynamic

__device__ inline void l_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct LEval {
    float dx, dy, raw, alpha;   // raw = opacity * exp(-sigma), before the clamp
    bool pass;
};
// identical instruction sequence to raster_shared.h's eval_splat (bit-identical alphas); A, B, C, L: the record's
// exponent form (common.h, write_splat_record)
__device__ __forceinline__ LEval l_eval(float gx, float gy, float A, float B, float C, float L, float px, float py) {
    LEval e;
    e.dx = gx - px;
    e.dy = gy - py;
    float s = __fmaf_rn(A * e.dx, e.dx, L);
    s = __fmaf_rn(C * e.dy, e.dy, s);
    s = __fmaf_rn(B * e.dx, e.dy, s);
    e.raw = __builtin_amdgcn_exp2f(s);
    e.alpha = fminf(L_ALPHA_MAX, e.raw);
    e.pass = !(s > L || e.alpha < L_ALPHA_MIN);
    return e;
}

struct LayerOut {
    float* render[NL];     // [C,H,W,CD] per layer (NULL when the layer is not requested)
    float* alphas[NL];     // [C,H,W]
    int32_t* last_ids[NL]; // [C,H,W]
};
struct LayerIn {
    const float* render_alphas[NL];
    const int32_t* last_ids[NL];
    const float* v_render[NL];  // NULL = zeros
    const float* v_alphas[NL];  // NULL = zeros
};

template <int CD>
struct LayerState {
    float T[LPPL];
    float acc[LPPL][CD];
    int last[LPPL];
    bool done[LPPL];
};

template <int CD>
__device__ __forceinline__ void layer_blend(LayerState<CD>& L, const LEval (&ev)[LPPL], const float (&col)[CD], int idx) {
#pragma unroll
    for (int k = 0; k < LPPL; ++k) {
        const bool pass = ev[k].pass && !L.done[k];
        const float nT = L.T[k] * (1.f - ev[k].alpha);
        const bool stop = pass && (nT <= L_T_STOP);
        L.done[k] = L.done[k] || stop;
        const bool blend = pass && !stop;
        const float w = blend ? ev[k].alpha * L.T[k] : 0.f;
#pragma unroll
        for (int c = 0; c < CD; ++c) L.acc[k][c] = __fmaf_rn(col[c], w, L.acc[k][c]);
        L.T[k] = blend ? nT : L.T[k];
        L.last[k] = blend ? idx : L.last[k];
    }
}

// outputs are [NL][C*H*W(*CD)]; layer_mask bit L set = layer L is wanted (layer 0 always)
template <int CD>
__global__ void __launch_bounds__(64 * LWAVES)
raster_layers_fwd_kernel(int n_tiles_total, int n_groups, int tile_w, int tile_h, int width, int height, int N, int Ns,
                         int layer_mask, const float* __restrict__ records, const float* __restrict__ backgrounds,
                         const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ tile_order,
                         const int32_t* __restrict__ flatten_ids, LayerOut out_ptrs) {
    constexpr int RS = (6 + CD + 3) & ~3;
    constexpr int RQ = RS / 4;
    __shared__ float4 slab[LWAVES][64][RQ];
    __shared__ int cls_of[LWAVES][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = wv & 1;
    int tile;
    if (tile_order) {  // heaviest-first schedule (see tile_scan_kernel); otherwise XCD-chunked raster order
        const int slot = blockIdx.x * 2 + (wv >> 1);
        if (slot >= (int)sched_slots((size_t)n_tiles_total)) return;
        tile = tile_order[slot];
        if (tile < 0) return;
        if (tile & SCHED_HEAVY) {  // a heavy tile fills 4 slots; this kernel always uses 2 waves per tile
            if (slot & 3) return;
            tile &= ~SCHED_HEAVY;
        }
    } else {
        const int group = xcd_chunked(blockIdx.x, n_groups);
        if (group >= n_groups) return;
        tile = group * 2 + (wv >> 1);
        if (tile >= n_tiles_total) return;
    }
    const int tiles_per_cam = tile_w * tile_h;
    const int cam = tile / tiles_per_cam;
    const int tl = tile - cam * tiles_per_cam;
    const int ty = tl / tile_w, tx = tl - ty * tile_w;
    const int pxi = tx * MOBGS_TILE + (lane & 15);
    const int pyi0 = ty * MOBGS_TILE + 8 * half + (lane >> 4);
    const float px = (float)pxi + 0.5f;
    float py[LPPL];
    bool outside[LPPL];
    LayerState<CD> L[NL];
#pragma unroll
    for (int k = 0; k < LPPL; ++k) {
        py[k] = (float)(pyi0 + 4 * k) + 0.5f;
        outside[k] = !(pxi < width && (pyi0 + 4 * k) < height);
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            L[l].T[k] = 1.f;
            L[l].last[k] = 0;
            L[l].done[k] = outside[k] || !((layer_mask >> l) & 1);
#pragma unroll
            for (int c = 0; c < CD; ++c) L[l].acc[k][c] = 0.f;
        }
    }
    const int s = __builtin_amdgcn_readfirstlane(tile_offsets[tile]);
    const int e = __builtin_amdgcn_readfirstlane(tile_offsets[tile + 1]);

    for (int b = s; b < e; b += 64) {
        const int n = min(64, e - b);
        l_fence();
        if (lane < n) {
            const int g = flatten_ids[b + lane];
            const float4* r = reinterpret_cast<const float4*>(records + (size_t)g * RS);
#pragma unroll
            for (int q = 0; q < RQ; ++q) slab[wv][lane][q] = r[q];
            cls_of[wv][lane] = ((g % N) < Ns) ? 1 : 2;
        }
        l_fence();
        bool all_done = false;
        for (int j = 0; j < n; ++j) {
            float rec[RS];
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const float4 v = slab[wv][j][q];
                rec[4 * q] = v.x;
                rec[4 * q + 1] = v.y;
                rec[4 * q + 2] = v.z;
                rec[4 * q + 3] = v.w;
            }
            const int cls = __builtin_amdgcn_readfirstlane(cls_of[wv][j]);
            LEval ev[LPPL];
#pragma unroll
            for (int k = 0; k < LPPL; ++k) ev[k] = l_eval(rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], px, py[k]);
            float col[CD];
#pragma unroll
            for (int c = 0; c < CD; ++c) col[c] = rec[6 + c];
            if (layer_mask & 1) layer_blend<CD>(L[0], ev, col, b + j);
            if (cls == 1) {
                if (layer_mask & 2) layer_blend<CD>(L[1], ev, col, b + j);
            } else {
                if (layer_mask & 4) layer_blend<CD>(L[2], ev, col, b + j);
            }
            bool live = false;
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int k = 0; k < LPPL; ++k) live = live || !L[l].done[k];
            if (__builtin_amdgcn_ballot_w64(live) == 0ull) {
                all_done = true;
                break;
            }
        }
        if (all_done) break;
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        if (!((layer_mask >> l) & 1)) continue;
#pragma unroll
        for (int k = 0; k < LPPL; ++k) {
            if (outside[k]) continue;
            const size_t pix = ((size_t)cam * height + (pyi0 + 4 * k)) * width + pxi;
            out_ptrs.alphas[l][pix] = 1.f - L[l].T[k];
            out_ptrs.last_ids[l][pix] = L[l].last[k];
            float* out = out_ptrs.render[l] + pix * CD;
#pragma unroll
            for (int c = 0; c < CD; ++c) {
                float v = L[l].acc[k][c];
                if (backgrounds) v = __fmaf_rn(L[l].T[k], backgrounds[cam * CD + c], v);
                out[c] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float l_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int NVP>
__device__ __forceinline__ void l_wave_reduce(float (&v)[NVP]) {
#pragma unroll
    for (int i = 0; i < NVP / 2; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + NVP / 2]), false,
                                                        false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < NVP / 4; ++i) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + NVP / 4]), false,
                                                        false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < NVP / 4; ++i) {
        float x = v[i];
        x = l_dpp_add<0xB1>(x);
        x = l_dpp_add<0x4E>(x);
        x = l_dpp_add<0x141>(x);
        x = l_dpp_add<0x128>(x);
        v[i] = x;
    }
}

__device__ __forceinline__ float l_wave_allreduce(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    v = l_dpp_add<0xB1>(v);
    v = l_dpp_add<0x4E>(v);
    v = l_dpp_add<0x141>(v);
    v = l_dpp_add<0x128>(v);
    return v;
}

template <int CD>
struct LayerBwd {
    float T[LPPL], Tf[LPPL], va[LPPL], bgdot[LPPL], behind[LPPL];
    float vo[LPPL][CD];
    int binf[LPPL];
};

// XY0 = true (layer 0 only): the position gradient is ALSO accumulated in (e0, e1) -- the reference's
// `viewspace_points.grad` is the means2d gradient of the combined render alone (:218-223, train.py:634-648)
template <int CD, int NVP, bool XY0>
__device__ __forceinline__ bool layer_grad(LayerBwd<CD>& B, const LEval (&ev)[LPPL], const float* rec, int idx,
                                           bool has_bg, float (&g)[NVP], float& e0, float& e1) {
    // this kernel still accumulates the conic-weighted terms per pair: conic and 1 / opacity back from the record's
    // exponent form (uniform per entry)
    struct { float ca, cb, cc, op; } co;
    record_conic_form(rec[2], rec[3], rec[4], rec[5], co.ca, co.cb, co.cc, co.op);
    const float inv_op = __builtin_amdgcn_exp2f(-rec[5]);
    bool any = false;
#pragma unroll
    for (int k = 0; k < LPPL; ++k) {
        const bool pass = ev[k].pass && (idx <= B.binf[k]);
        any = any || pass;
        if (!pass) continue;
        const float alpha = ev[k].alpha;
        const float om = 1.f - alpha;
        const float ra = __builtin_amdgcn_rcpf(om);   // (no Newton step: raster.hip, blend_bwd)
        B.T[k] *= ra;
        const float fac = alpha * B.T[k];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < CD; ++c) {
            g[6 + c] = __fmaf_rn(fac, B.vo[k][c], g[6 + c]);
            dot = __fmaf_rn(rec[6 + c], B.vo[k][c], dot);
        }
        float v_alpha = __fmaf_rn(B.T[k], dot, -ra * B.behind[k]);
        v_alpha += B.Tf[k] * ra * B.va[k];
        if (has_bg) v_alpha -= B.Tf[k] * ra * B.bgdot[k];
        const float ov = ev[k].raw;
        if (ov <= L_ALPHA_MAX) {
            const float v_sigma = -ov * v_alpha;
            const float dx = ev[k].dx, dy = ev[k].dy;
            const float gx = v_sigma * (co.ca * dx + co.cb * dy), gy = v_sigma * (co.cb * dx + co.cc * dy);
            g[0] += gx;
            g[1] += gy;
            if (XY0) {
                e0 += gx;
                e1 += gy;
            }
            g[2] = __fmaf_rn(0.5f * v_sigma * dx, dx, g[2]);
            g[3] = __fmaf_rn(v_sigma * dx, dy, g[3]);
            g[4] = __fmaf_rn(0.5f * v_sigma * dy, dy, g[4]);
            g[5] = __fmaf_rn(ov * inv_op, v_alpha, g[5]);   // visibility = raw / opacity
        }
        B.behind[k] = __fmaf_rn(fac, dot, B.behind[k]);
    }
    return any;
}

// grad_slots [I][2 halves][RS]; v_render / v_alphas / render_alphas / last_ids are [NL][C*H*W(*CD)]
template <int CD>
__global__ void __launch_bounds__(64 * LWAVES)
raster_layers_bwd_kernel(int n_tiles_total, int n_groups, int tile_w, int tile_h, int width, int height, int N, int Ns,
                         int layer_mask, const float* __restrict__ records, const float* __restrict__ backgrounds,
                         const int32_t* __restrict__ radii, const int32_t* __restrict__ cum_tiles,
                         const int32_t* __restrict__ keep_scan, const int32_t* __restrict__ tile_offsets,
                         const int32_t* __restrict__ tile_order, const int32_t* __restrict__ flatten_ids, LayerIn in_ptrs,
                         float* __restrict__ grad_slots, float* __restrict__ grad_xy0) {
    constexpr int RS = (6 + CD + 3) & ~3;
    constexpr int RQ = RS / 4;
    constexpr int NV = 6 + CD;
    constexpr int NVP = NV <= 8 ? 8 : (NV <= 16 ? 16 : 32);
    __shared__ float4 slab[LWAVES][64][RQ];
    __shared__ int slot_of[LWAVES][64];
    __shared__ int cls_of[LWAVES][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = wv & 1;
    int tile;
    if (tile_order) {  // heaviest-first schedule (see tile_scan_kernel); otherwise XCD-chunked raster order
        const int slot = blockIdx.x * 2 + (wv >> 1);
        if (slot >= (int)sched_slots((size_t)n_tiles_total)) return;
        tile = tile_order[slot];
        if (tile < 0) return;
        if (tile & SCHED_HEAVY) {  // a heavy tile fills 4 slots; this kernel always uses 2 waves per tile
            if (slot & 3) return;
            tile &= ~SCHED_HEAVY;
        }
    } else {
        const int group = xcd_chunked(blockIdx.x, n_groups);
        if (group >= n_groups) return;
        tile = group * 2 + (wv >> 1);
        if (tile >= n_tiles_total) return;
    }
    const int tiles_per_cam = tile_w * tile_h;
    const int cam = tile / tiles_per_cam;
    const int tl = tile - cam * tiles_per_cam;
    const int ty = tl / tile_w, tx = tl - ty * tile_w;
    const int pxi = tx * MOBGS_TILE + (lane & 15);
    const int pyi0 = ty * MOBGS_TILE + 8 * half + (lane >> 4);
    const float px = (float)pxi + 0.5f;
    const int s = __builtin_amdgcn_readfirstlane(tile_offsets[tile]);
    const int e = __builtin_amdgcn_readfirstlane(tile_offsets[tile + 1]);
    if (e <= s) return;
    const bool has_bg = backgrounds != nullptr;

    float py[LPPL];
    LayerBwd<CD> B[NL];
    int top = -1;
#pragma unroll
    for (int k = 0; k < LPPL; ++k) {
        const int pyi = pyi0 + 4 * k;
        py[k] = (float)pyi + 0.5f;
        const bool inside = pxi < width && pyi < height;
        const size_t pix = ((size_t)cam * height + pyi) * width + pxi;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            B[l].binf[k] = -1;
            B[l].Tf[k] = 1.f;
            B[l].va[k] = 0.f;
            B[l].bgdot[k] = 0.f;
            B[l].behind[k] = 0.f;
#pragma unroll
            for (int c = 0; c < CD; ++c) B[l].vo[k][c] = 0.f;
            if (inside && ((layer_mask >> l) & 1)) {
                B[l].binf[k] = in_ptrs.last_ids[l][pix];
                B[l].Tf[k] = 1.f - in_ptrs.render_alphas[l][pix];
                B[l].va[k] = in_ptrs.v_alphas[l] ? in_ptrs.v_alphas[l][pix] : 0.f;
                if (in_ptrs.v_render[l]) {
                    const float* vr = in_ptrs.v_render[l] + pix * CD;
#pragma unroll
                    for (int c = 0; c < CD; ++c) B[l].vo[k][c] = vr[c];
                }
                if (has_bg) {
#pragma unroll
                    for (int c = 0; c < CD; ++c)
                        B[l].bgdot[k] = __fmaf_rn(backgrounds[cam * CD + c], B[l].vo[k][c], B[l].bgdot[k]);
                }
                top = max(top, B[l].binf[k]);
            }
            B[l].T[k] = B[l].Tf[k];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) top = max(top, __shfl_xor(top, off, 64));
    top = min(top, e - 1);

    for (int hi = top; hi >= s; hi -= 64) {
        const int n = min(64, hi - s + 1);
        l_fence();
        if (lane < n) {
            const int g = flatten_ids[hi - lane];
            const float4* r = reinterpret_cast<const float4*>(records + (size_t)g * RS);
            const float4 r0 = r[0];
            slab[wv][lane][0] = r0;
#pragma unroll
            for (int q = 1; q < RQ; ++q) slab[wv][lane][q] = r[q];
            const TileRect tr = tile_rect(r0.x, r0.y, radii[g], tile_w, tile_h);
            slot_of[wv][lane] = keep_index(keep_scan, cum_tiles[g] + (ty - tr.y0) * (tr.x1 - tr.x0) + (tx - tr.x0));
            cls_of[wv][lane] = ((g % N) < Ns) ? 1 : 2;
        }
        l_fence();
        for (int j = 0; j < n; ++j) {
            const int idx = hi - j;
            float rec[RS];
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const float4 v = slab[wv][j][q];
                rec[4 * q] = v.x;
                rec[4 * q + 1] = v.y;
                rec[4 * q + 2] = v.z;
                rec[4 * q + 3] = v.w;
            }
            const int cls = __builtin_amdgcn_readfirstlane(cls_of[wv][j]);
            LEval ev[LPPL];
#pragma unroll
            for (int k = 0; k < LPPL; ++k) ev[k] = l_eval(rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], px, py[k]);
            float g[NVP];
#pragma unroll
            for (int i = 0; i < NVP; ++i) g[i] = 0.f;
            float e0 = 0.f, e1 = 0.f, dummy0 = 0.f, dummy1 = 0.f;
            bool any = false;
            if (layer_mask & 1) any = layer_grad<CD, NVP, true>(B[0], ev, rec, idx, has_bg, g, e0, e1);
            if (cls == 1) {
                if (layer_mask & 2) any = layer_grad<CD, NVP, false>(B[1], ev, rec, idx, has_bg, g, dummy0, dummy1) || any;
            } else {
                if (layer_mask & 4) any = layer_grad<CD, NVP, false>(B[2], ev, rec, idx, has_bg, g, dummy0, dummy1) || any;
            }
            if (__builtin_amdgcn_ballot_w64(any) == 0ull) continue;
            l_wave_reduce<NVP>(g);
            e0 = l_wave_allreduce(e0);
            e1 = l_wave_allreduce(e1);
            if (lane == 0)
                *reinterpret_cast<float2*>(grad_xy0 + ((size_t)slot_of[wv][j] * 2 + half) * 2) = make_float2(e0, e1);
            constexpr int Q = NVP / 4;
            const int base = (lane >> 5) * (NVP / 2) + ((lane >> 4) & 1) * Q;
            if ((lane & 15) == 0 && base < RS) {
                float* dst = grad_slots + ((size_t)slot_of[wv][j] * 2 + half) * RS + base;
#pragma unroll
                for (int q = 0; q < Q; q += 4)
                    if (base + q < RS) *reinterpret_cast<float4*>(dst + q) = make_float4(g[q], g[q + 1], g[q + 2], g[q + 3]);
            }
        }
    }
}

// per-splat sum of its 2 * n_slots half-tile records
__global__ void __launch_bounds__(256)
layers_slot_reduce_kernel(int n_gauss, int channels, int has_extra, int stride, const int32_t* __restrict__ cum_tiles,
                          const int32_t* __restrict__ keep_scan, const float* __restrict__ grad_slots,
                          const float* __restrict__ grad_xy0, float* __restrict__ v_means2d_l0,
                          float* __restrict__ v_means2d, float* __restrict__ v_conics,
                          float* __restrict__ v_opacities, float* __restrict__ v_colors, float* __restrict__ v_extra,
                          const int32_t* __restrict__ tiles_per_gauss, const int32_t* __restrict__ lists_total) {
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / 16;
    const int comp = threadIdx.x % 16;
    if (gid >= n_gauss) return;
    // (with a caller-chosen enumeration order of the intersections only start + count is the end: mobgs_hip.h, enum_order)
    const int end_box = tiles_per_gauss ? cum_tiles[gid] + tiles_per_gauss[gid] : cum_tiles[gid + 1];
    // lists_total = tile_offsets[n_tiles]: 0 when an arena overflowed and the lists were emptied (speculative binning without
    // a host in the loop, e.g. a HIP-graph replay) -- keep_scan and the slot ranges are then NOT to be trusted: every sum is 0
    const bool empty = lists_total && *lists_total == 0;
    const int a = empty ? 0 : 2 * keep_index(keep_scan, cum_tiles[gid]), b = empty ? 0 : 2 * keep_index(keep_scan, end_box);
    float acc = 0.f;
    if (comp < stride) {
        const float* p = grad_slots + (size_t)a * stride + comp;
        for (int k = a; k < b; ++k, p += stride) acc += *p;
    }
    const size_t g = (size_t)gid;
    if (comp < 2) {
        v_means2d[2 * g + comp] = acc;
        float a0 = 0.f;
        const float* q = grad_xy0 + (size_t)a * 2 + comp;
        for (int k = a; k < b; ++k, q += 2) a0 += *q;
        v_means2d_l0[2 * g + comp] = a0;
    } else if (comp < 5)
        v_conics[3 * g + (comp - 2)] = acc;
    else if (comp == 5)
        v_opacities[g] = acc;
    else if (comp - 6 < channels)
        v_colors[g * channels + (comp - 6)] = acc;
    else if (has_extra && comp - 6 == channels)
        v_extra[g] = acc;
}

}  // namespace mobgs

using namespace mobgs;

extern "C" {

int mobgs_raster_layers_fwd(int C, int N, int Ns, int layer_mask, int channels_total, int width, int height,
                            const float* records, const float* backgrounds, const int32_t* tile_offsets,
                            const int32_t* tile_order, const int32_t* flatten_ids, float* const* render3_host, float* const* alphas3_host,
                            int32_t* const* last_ids3_host, void* stream) {
    if (C <= 0 || N < 0 || Ns < 0 || Ns > N || channels_total != 10 || !(layer_mask & 7)) {
        set_error("mobgs_raster_layers_fwd: unsupported arguments (C=%d N=%d Ns=%d D=%d mask=%d)", C, N, Ns,
                  channels_total, layer_mask);
        return MOBGS_E_UNSUPPORTED;
    }
    LayerOut o;
    for (int l = 0; l < NL; ++l) {
        o.render[l] = render3_host[l];
        o.alphas[l] = alphas3_host[l];
        o.last_ids[l] = last_ids3_host[l];
        if (((layer_mask >> l) & 1) && !(o.render[l] && o.alphas[l] && o.last_ids[l])) {
            set_error("mobgs_raster_layers_fwd: layer %d requested but its output pointers are NULL", l);
            return MOBGS_E_INVALID;
        }
    }
    const int tile_w = (width + MOBGS_TILE - 1) / MOBGS_TILE, tile_h = (height + MOBGS_TILE - 1) / MOBGS_TILE;
    const int nt = C * tile_w * tile_h;
    const int n_groups = (nt + 1) / 2;
    const int grid = tile_order ? (int)((sched_slots((size_t)nt) + 1) / 2) : ((n_groups + 7) / 8) * 8;
    hipLaunchKernelGGL(raster_layers_fwd_kernel<10>, dim3(grid), dim3(64 * LWAVES), 0, (hipStream_t)stream, nt, n_groups,
                       tile_w, tile_h, width, height, N, Ns, layer_mask, records, backgrounds, tile_offsets, tile_order,
                       flatten_ids, o);
    return check_launch("raster_layers_fwd_kernel");
}

int mobgs_raster_layers_bwd(int C, int N, int Ns, int layer_mask, int channels, int has_extra, int width, int height,
                            const float* records, const float* backgrounds, const int32_t* radii,
                            const int32_t* cum_tiles, const int32_t* keep_scan, const int32_t* tile_offsets,
                            const int32_t* tile_order, const int32_t* flatten_ids,
                            const float* const* render_alphas3_host,
                            const int32_t* const* last_ids3_host, const float* const* v_render3_host,
                            const float* const* v_alphas3_host, float* grad_slots, float* grad_xy0,
                            float* v_means2d_layer0, float* v_means2d, float* v_conics, float* v_opacities,
                            float* v_colors, float* v_extra, const int32_t* tiles_per_gauss, void* stream) {
    const int D = channels + (has_extra ? 1 : 0);
    if (C <= 0 || N < 0 || D != 10 || !(layer_mask & 7)) {
        set_error("mobgs_raster_layers_bwd: unsupported arguments");
        return MOBGS_E_UNSUPPORTED;
    }
    LayerIn in;
    for (int l = 0; l < NL; ++l) {
        in.render_alphas[l] = render_alphas3_host[l];
        in.last_ids[l] = last_ids3_host[l];
        in.v_render[l] = v_render3_host[l];
        in.v_alphas[l] = v_alphas3_host[l];
        if (((layer_mask >> l) & 1) && !(in.render_alphas[l] && in.last_ids[l])) {
            set_error("mobgs_raster_layers_bwd: layer %d requested but its saved tensors are NULL", l);
            return MOBGS_E_INVALID;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const int tile_w = (width + MOBGS_TILE - 1) / MOBGS_TILE, tile_h = (height + MOBGS_TILE - 1) / MOBGS_TILE;
    const int nt = C * tile_w * tile_h;
    const int n_groups = (nt + 1) / 2;
    const int grid = tile_order ? (int)((sched_slots((size_t)nt) + 1) / 2) : ((n_groups + 7) / 8) * 8;
    hipLaunchKernelGGL(raster_layers_bwd_kernel<10>, dim3(grid), dim3(64 * LWAVES), 0, st, nt, n_groups, tile_w, tile_h,
                       width, height, N, Ns, layer_mask, records, backgrounds, radii, cum_tiles, keep_scan, tile_offsets,
                       tile_order, flatten_ids, in, grad_slots, grad_xy0);
    const int n = C * N;
    if (n > 0)
        hipLaunchKernelGGL(layers_slot_reduce_kernel, dim3((int)(((size_t)n * 16 + 255) / 256)), dim3(256), 0, st, n,
                           channels, has_extra, record_stride(D), cum_tiles, keep_scan, grad_slots, grad_xy0,
                           v_means2d_layer0, v_means2d, v_conics, v_opacities, v_colors, v_extra, tiles_per_gauss,
                           tile_offsets ? tile_offsets + nt : nullptr);
    return check_launch("raster_layers_bwd_kernel");
}

}  // extern "C"
