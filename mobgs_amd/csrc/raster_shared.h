// Pieces shared by the compositing translation units (raster.hip, raster_bwd_mfma.hip): constants of gsplat's blend
// rule, the tile schedule lookup, the class selector of the restricted passes and the per-(pixel, splat) evaluation --
// ONE instruction sequence for forward and backward, so both see bit-identical alphas.
#pragma once
#include "common.h"

namespace mobgs {

constexpr float ALPHA_MIN = 1.f / 255.f;
constexpr float ALPHA_MAX = 0.999f;
constexpr float T_STOP = 1e-4f;
constexpr int PPL = 4;            // pixels per lane
constexpr int TILES_PER_WG = 4;   // waves per workgroup, each on its own tile

// Which tile this wave works on: with a schedule, workgroups take tiles in the given order (heaviest first);
// without, tiles are taken in XCD-chunked raster order.  -1: nothing to do.  With a schedule the value may carry
// SCHED_HEAVY: the 4 waves of the workgroup then share that ONE tile, wave w taking its 8x8 quadrant w.
__device__ inline int scheduled_tile(const int32_t* tile_order, int n_groups, int n_tiles_total, int wv) {
    if (tile_order) {
        const int slot = blockIdx.x * TILES_PER_WG + wv;
        return slot < (int)sched_slots((size_t)n_tiles_total) ? tile_order[slot] : -1;
    }
    const int group = xcd_chunked(blockIdx.x, n_groups);
    if (group >= n_groups) return -1;
    const int tile = group * TILES_PER_WG + wv;
    return tile < n_tiles_total ? tile : -1;
}

__device__ inline void wave_lds_fence() {
    // LDS operations of one wave execute in issue order; only the compiler has to be told
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Optional restriction of a compositing pass to one class of splats: sel = 1 keeps flat ids with (id % N) < Ns
// ("static"), sel = 2 the others ("dynamic"), 0 = no restriction.  The pass walks the SAME per-tile lists and list
// indices as the unrestricted one; entries of the other class are dropped when a batch is staged, so they cost a
// class test per batch, not a trip through the blend loop.
struct ClassSel {
    int sel, N, Ns;
    int all_reach;  // testing aid (MobgsTuning.quadrant_culling = 0): treat every quadrant as reachable
    // backward passes only (MobgsTuning.gate_zero_cotangent): a device word that is non-zero iff some cotangent of this
    // pass is; when it reads 0 the kernel returns at once -- no record is written, any_record stays 0 and the slot
    // reduction then writes exact zeros without reading a slot
    const int32_t* gate = nullptr;
    // backward passes only (MobgsTuning.static_rows, round 6): rows (flat id % set_n) < static_rows are MoBGS's STATIC
    // splats -- colour features cat(f_dc, 0.0 * f_t) (/root/reference/scene/gaussian_model.py:244-246), flow channels
    // identically zero -- whose dead channels the caller neither fills nor wants a gradient for (dead_channels<CD>)
    int static_rows = 0, set_n = 1;
    // backward passes only (MobgsTuning.cover_slots, round 6): the caller did NOT zero-fill grad_slots -- the kernel writes
    // the slot of EVERY entry of its lists (zeros where no pixel blended the splat)
    int cover = 0;
    __device__ __forceinline__ bool gated_off() const { return gate && *gate == 0; }
    __device__ __forceinline__ bool keeps(int flat_id) const { return sel == 0 || (((flat_id % N) < Ns) == (sel == 1)); }
    __device__ __forceinline__ bool static_row(int flat_id) const {
        return static_rows > 0 && (flat_id % set_n) < static_rows;
    }
};

// Colour channels [DEAD_FIRST, DEAD_FIRST + dead_channels<CD>) of a STATIC splat's record are structurally zero in the
// two passes that carry MoBGS's feature layout: CD = 10 = [f_dc(6) | t f_t(3) | depth] (render(),
// /root/reference/gaussian_renderer/__init__.py:125,201-217: the static rows hold 0.0 * f_t) and CD = 12 = [f_dc(6) |
// t f_t(3) | flow(2) | depth] (get_flow(), :436-476: a static splat projects to the same pixel at both exposures, its
// flow is x - x = 0).  fma(0, w, acc) = acc: leaving those FMAs out changes no bit of any other sum; the gradient of the
// dead channels themselves is multiplied by 0.0 downstream (prep_shared.h: g_s_ft = 0.0f * v_col) and is written as 0.
constexpr int DEAD_FIRST = 6;
template <int CD>
constexpr int dead_channels() { return CD == 10 ? 3 : (CD == 12 ? 5 : 0); }
// ... and that is CHECKED per staged entry (the record's quarters pass through registers then: 3 / 5 compares per 64
// entries): an entry of a static row whose dead channels are not all +-0 simply takes the full blend body.
// -> quarter q (floats 4 q .. 4 q + 3 of the record) holds no non-zero dead channel
template <int CD>
__device__ __forceinline__ bool quarter_dead_zero(int q, const float4& v) {
    if constexpr (CD == 10) return q != 3 || (v.x == 0.f && v.y == 0.f && v.z == 0.f);                 // floats 12..14
    if constexpr (CD == 12)
        return q == 3 ? (v.x == 0.f && v.y == 0.f && v.z == 0.f && v.w == 0.f) : (q != 4 || v.x == 0.f);  // 12..16
    return false;
}

// offsets / unclamped alpha / alpha of one splat at one pixel from the record's exponent form (common.h,
// write_splat_record): raw = opacity * exp(-sigma) = exp2(A dx^2 + C dy^2 + B dx dy + L); the same instruction
// sequence in every forward and backward kernel, so all of them see bit-identical alphas.
// pass = !(sigma < 0 || alpha < 1/255): sigma < 0 <=> the exponent exceeds L.
struct Eval {
    float dx, dy, raw, alpha;
    bool pass;
};
__device__ __forceinline__ Eval eval_splat(float gx, float gy, float A, float B, float C, float L, float px, float py) {
    Eval e;
    e.dx = gx - px;
    e.dy = gy - py;
    float s = __fmaf_rn(A * e.dx, e.dx, L);
    s = __fmaf_rn(C * e.dy, e.dy, s);
    s = __fmaf_rn(B * e.dx, e.dy, s);
    e.raw = __builtin_amdgcn_exp2f(s);
    e.alpha = fminf(ALPHA_MAX, e.raw);
    e.pass = !(s > L || e.alpha < ALPHA_MIN);
    return e;
}
// conic and opacity of a staged record head {x, y, A, B | C, L, ..} for the reach tests
struct ConicOp {
    float ca, cb, cc, op;
};
__device__ __forceinline__ ConicOp head_conic(const float4& r0, const float4& r1) {
    ConicOp c;
    record_conic_form(r0.z, r0.w, r1.x, r1.y, c.ca, c.cb, c.cc, c.op);
    return c;
}
__device__ __forceinline__ unsigned quadrant_reach_mask_rec(const float4& r0, const float4& r1, int tx, int ty) {
    const ConicOp c = head_conic(r0, r1);
    return quadrant_reach_mask(r0.x, r0.y, c.ca, c.cb, c.cc, c.op, tx, ty);
}

}  // namespace mobgs
