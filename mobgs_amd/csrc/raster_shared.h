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
    __device__ __forceinline__ bool gated_off() const { return gate && *gate == 0; }
    __device__ __forceinline__ bool keeps(int flat_id) const { return sel == 0 || (((flat_id % N) < Ns) == (sel == 1)); }
};

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
