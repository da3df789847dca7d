// K6 / K7: front-to-back alpha compositing of depth-sorted splats, forward and backward, for gfx950.
//
// Semantics follow gsplat v1.4.0 rasterize_to_pixels_{fwd,bwd}.cu [upstream, SURVEY.md Appendix A.3/A.4],
// reached from the reference through rasterization() at
// /root/reference/gaussian_renderer/__init__.py:143,163,201,236,255,274,379,437,456,473,538.
//
// MI355X mapping (not upstream's 256-thread block per tile):
//   * one 64-lane wavefront owns one 16x16 tile; every lane carries 4 pixels -- the same (lane&7, lane>>3)
//     position in each of the four 8x8 QUADRANTS of the tile -- so per-splat uniform work (record fetch, loop
//     control, early-out votes) is amortised over 256 pixel evaluations and no workgroup barrier exists anywhere
//     in the kernels.  Quadrants rather than 16x4 strips: a splat's footprint is a compact blob, so it usually
//     misses whole quadrants (their code is skipped by a wave-level branch) and fills the ones it hits;
//   * splats are staged 64 at a time (lane = splat) as packed 16-float records
//     {x, y, conic a b c, opacity, colour[..]} into a per-wave LDS slab and re-read as broadcasts;
//   * while a batch is staged every lane also works out which of the four quadrants ITS splat can reach at all
//     (quadrant_reach_mask, common.h); evaluation and blend of the others are skipped by scalar branches -- ~40 % of
//     the (entry, quadrant) pairs of a typical list, bit-identical results (MobgsTuning.quadrant_culling = 0 turns it
//     off for the test that proves it);
//   * backward reduces the 6+D per-splat gradient components across the wave with a halving butterfly
//     (log-depth, D+6 -> 1 value per lane) and writes ONE 64-byte gradient record per (tile, splat) into a
//     slot owned by that intersection; a second streaming kernel sums each splat's contiguous slots.
//     No floating-point atomics => bit-reproducible gradients.
#include <type_traits>

#include "common.h"
#include "decoder_shared.h"
#include "raster_shared.h"

namespace mobgs {

// raster_bwd_mfma.hip: the backward compositor with the gradient sums on the matrix pipe (<= 10 total channels)
bool raster_bwd_mfma_launch(int mode, int D, bool filter, int grid, hipStream_t st, int nt, int n_groups, int tile_w, int tile_h,
                            int width, int height, const float* records, const float* backgrounds,
                            const int32_t* radii, const int32_t* cum_tiles, const int32_t* keep_scan,
                            const int32_t* tile_offsets, const int32_t* flatten_ids, const float* render_alphas,
                            const int32_t* last_ids, const float* v_render, const float* v_alphas, float* grad_slots,
                            const int32_t* tile_order, ClassSel cls, const uint8_t* isect_reach, int32_t* any_record);

// ---------------------------------------------------------------------------------------------------
// pack: gather the per-splat inputs of the compositor into one aligned record
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pack_records_kernel(int N, int channels, int stride, const float* __restrict__ means2d,
                    const float* __restrict__ conics, const float* __restrict__ colors, int colors_per_camera,
                    const float* __restrict__ opacities, int opac_per_camera, const float* __restrict__ extra,
                    const int32_t* __restrict__ radii, float* __restrict__ records) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (i >= N) return;
    const size_t o = (size_t)c * N + i;
    if (radii[o] <= 0) return;  // culled splats are never referenced by a tile list
    const float2 m = reinterpret_cast<const float2*>(means2d)[o];
    write_splat_record(records + o * stride, m.x, m.y, conics[3 * o], conics[3 * o + 1], conics[3 * o + 2],
                       opacities[opac_per_camera ? o : (size_t)i], colors + (colors_per_camera ? o : (size_t)i) * channels,
                       channels, extra != nullptr, extra ? extra[o] : 0.f);
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// The Sandwich decoder as the EPILOGUE of the forward compositor (round 5; 10 total channels = 9 features + depth, pinhole
// rays generated in registers): the pixel's composited features are in registers when its walk ends, so the decoded
// colour and the expected depth leave the same kernel -- no decoder launch, no re-read of the 55-MB feature image
// (decoder_fwd: 18 us; the image itself is still written: the backward pass reads it).  Same instruction sequence as
// decoder_fwd_kernel (decoder_shared.h): bit-identical images.  rgb == NULL: off.
struct DecodeEpi {
    const float *intr, *c2w, *w1, *w2;   // [fx, fy, cx, cy] (+ intr_stride per camera), 3x4 pose (+ c2w_stride), weights
    float *rgb, *depth;                  // [C,3,H,W], [C,H,W]
    int intr_stride, c2w_stride;
};
// NPX pixels of one lane (pxi[k], pyi[k]; `inside` bit k = the pixel exists).  ROW4: the four pixels are consecutive in x
// starting at a multiple of 4 (the block-walk lane layout): one 16-byte store per output plane when the image width allows.
template <int NPX, bool ROW4>
__device__ __forceinline__ void decode_epilogue(const DecodeEpi& d, const RayCam& rc, int cam, const int (&pxi)[NPX],
                                                const int (&pyi)[NPX], unsigned inside, int width, int height,
                                                const float (&f)[NPX][10], const float (&alpha)[NPX]) {
    const int P = width * height;
    float r[NPX][6], out[NPX][3];
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        float loc[2], inv_n;
        pixel_ray_xy(rc, pxi[k], pyi[k], r[k], loc, inv_n);
    }
    sandwich_forward_n<NPX>(d.w1, d.w2, f, r, out);
    float* rgb = d.rgb + (size_t)cam * 3 * P;
    float* dep = d.depth + (size_t)cam * P;
    if (ROW4 && NPX == 4 && inside == 0xFu && (width & 3) == 0) {
        const int p = pyi[0] * width + pxi[0];
#pragma unroll
        for (int o = 0; o < 3; ++o)
            *reinterpret_cast<float4*>(rgb + (size_t)o * P + p) = make_float4(out[0][o], out[1][o], out[2][o], out[3][o]);
        *reinterpret_cast<float4*>(dep + p) = make_float4(f[0][9] / fmaxf(alpha[0], 1e-10f), f[1][9] / fmaxf(alpha[1], 1e-10f),
                                                          f[2][9] / fmaxf(alpha[2], 1e-10f), f[3][9] / fmaxf(alpha[3], 1e-10f));
        return;
    }
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        if (!((inside >> k) & 1u)) continue;
        const int p = pyi[k] * width + pxi[k];
#pragma unroll
        for (int o = 0; o < 3; ++o) rgb[(size_t)o * P + p] = out[k][o];
        dep[p] = f[k][9] / fmaxf(alpha[k], 1e-10f);
    }
}

// One wave composites NP pixels per lane of `tile` front to back: NP = 4 -> the whole 16x16 tile (pixel k of a lane
// lies in quadrant k), NP = 1 -> only the 8x8 quadrant `quad` (heavy tiles: 4 waves share the list walk, which cuts
// the critical path of a long list ~2.6x; every pixel sees exactly the same arithmetic either way).
template <int CD, int NP, bool FILTER, bool DECODE = false>
__device__ __forceinline__ void composite_fwd(int tile, int quad, int wv, int lane,
                                              float4 (*slab)[64][((6 + CD + 3) & ~3) / 4], int (*idx_of)[64],
                                              unsigned (*reach_of)[64], ClassSel cls, int tile_w, int tile_h,
                                              int width, int height, const float* __restrict__ records,
                                              const float* __restrict__ backgrounds,
                                              const int32_t* __restrict__ tile_offsets,
                                              const int32_t* __restrict__ flatten_ids, float* __restrict__ render,
                                              float* __restrict__ alphas, int32_t* __restrict__ last_ids,
                                              uint8_t* __restrict__ isect_reach, const DecodeEpi* dec = nullptr) {
    constexpr int RS = (6 + CD + 3) & ~3;
    constexpr int RQ = RS / 4;
    constexpr int PPL = NP;
    const int tiles_per_cam = tile_w * tile_h;
    const int cam = tile / tiles_per_cam;
    const int tl = tile - cam * tiles_per_cam;
    const int ty = tl / tile_w, tx = tl - ty * tile_w;
    int pxi[PPL], pyi[PPL];
    float px[PPL], py[PPL];
    // T < 0 marks a pixel that takes no more splats (transmittance exhausted, or outside the image); |T| is its
    // transmittance.  The flag rides in the sign so that "finished" costs no register and no test of its own: a
    // negative T makes the candidate transmittance negative, which the stop test below already rejects.
    float T[PPL];
    float acc[PPL][CD];
    int last[PPL];
    unsigned alive = 0u;  // wave-uniform: pixel slots k that still have an unfinished pixel in some lane
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int qd = NP == 4 ? k : quad;
        pxi[k] = tx * MOBGS_TILE + 8 * (qd & 1) + (lane & 7);
        pyi[k] = ty * MOBGS_TILE + 8 * (qd >> 1) + (lane >> 3);
        px[k] = (float)pxi[k] + 0.5f;
        py[k] = (float)pyi[k] + 0.5f;
        const bool inside = pxi[k] < width && pyi[k] < height;
        T[k] = inside ? 1.f : -1.f;
        last[k] = 0;
        if (__builtin_amdgcn_ballot_w64(inside) != 0ull) alive |= 1u << k;
#pragma unroll
        for (int c = 0; c < CD; ++c) acc[k][c] = 0.f;
    }

    const int s = __builtin_amdgcn_readfirstlane(tile_offsets[tile]);
    const int e = __builtin_amdgcn_readfirstlane(tile_offsets[tile + 1]);

    // software pipeline: the records of batch b+1 are fetched into registers while batch b is blended
    constexpr int PQ = RQ < 2 ? RQ : 2;  // prefetched quarters: position, conic, opacity (and 2 colours)
    float4 pre[PQ];
#pragma unroll
    for (int q = 0; q < PQ; ++q) pre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    bool pre_keep = false;  // FILTER: the prefetched entry belongs to the wanted class
    int pre_g = 0;
    if (s + lane < e) {
        pre_g = flatten_ids[s + lane];
        pre_keep = !FILTER || cls.keeps(pre_g);
        if (pre_keep) {
            const float4* r = reinterpret_cast<const float4*>(records + (size_t)pre_g * RS);
#pragma unroll
            for (int q = 0; q < PQ; ++q) pre[q] = r[q];
        }
    }
    for (int b = s; b < e && alive != 0u; b += 64) {
        int n = min(64, e - b);
        wave_lds_fence();
        {
            // the colour quarters of the record were not prefetched (8 registers fewer across the blend loop): fetch
            // them now, the reach computation below covers their latency
            float4 rest[RQ > PQ ? RQ - PQ : 1];
            if (RQ > PQ && pre_keep) {
                const float4* r = reinterpret_cast<const float4*>(records + (size_t)pre_g * RS);
#pragma unroll
                for (int q = PQ; q < RQ; ++q) rest[q - PQ] = r[q];
            }
            // which 8x8 quadrants of the tile the splat can reach at all (lane = splat: one test per entry, not per
            // pixel): ~40 % of the (entry, quadrant) pairs of a typical list are out of reach and are never evaluated
            unsigned reach = cls.all_reach ? 0xFu
                                           : quadrant_reach_mask_rec(pre[0], pre[1], tx, ty);
            // kept for the backward pass (it walks the same lists): one byte per list entry.  (Heavy tiles: all four
            // quadrant waves store the same byte -- any of them may leave the walk first.)
            if (isect_reach && b + lane < e && (!FILTER || pre_keep)) isect_reach[b + lane] = (uint8_t)reach;
            if (NP == 1) reach = (reach >> quad) & 1u;
            int pos = lane;
            bool keep = true;
            if (FILTER) {  // stage only the wanted class, compacted, remembering each entry's list index
                const unsigned long long km = __builtin_amdgcn_ballot_w64(pre_keep);
                n = __builtin_popcountll(km);
                pos = __builtin_popcountll(km & ((1ull << lane) - 1ull));
                keep = pre_keep;
            }
            if (keep) {
#pragma unroll
                for (int q = 0; q < PQ; ++q) slab[wv][pos][q] = pre[q];
#pragma unroll
                for (int q = PQ; q < RQ; ++q) slab[wv][pos][q] = rest[q - PQ];
                reach_of[wv][pos] = reach;
                if (FILTER) idx_of[wv][pos] = b + lane;
            }
        }
        wave_lds_fence();
        pre_keep = false;
        if (b + 64 + lane < e) {
            pre_g = flatten_ids[b + 64 + lane];
            pre_keep = !FILTER || cls.keeps(pre_g);
            if (pre_keep) {
                const float4* r = reinterpret_cast<const float4*>(records + (size_t)pre_g * RS);
#pragma unroll
                for (int q = 0; q < PQ; ++q) pre[q] = r[q];
            }
        }
        // (wave-uniform 64-bit reach masks + scalar bit tests per entry, as the backward walk uses, measured 16 us
        // SLOWER here than reading each entry's mask out of its lane)
        const unsigned reach_lane = reach_of[wv][lane];  // lane j: the mask of staged entry j
        for (int j = 0; j < n; ++j) {
            const unsigned todo = (unsigned)__builtin_amdgcn_readlane((int)reach_lane, j) & alive;
            if (todo == 0u) continue;
            const int list_idx = FILTER ? idx_of[wv][j] : b + j;
            float rec[RS];
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const float4 v = slab[wv][j][q];
                rec[4 * q] = v.x;
                rec[4 * q + 1] = v.y;
                rec[4 * q + 2] = v.z;
                rec[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                if (!((todo >> k) & 1u)) continue;  // wave-uniform
                const Eval ev = eval_splat(rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], px[k], py[k]);
                const float nT = T[k] * (1.f - ev.alpha);
                const bool blend = ev.pass && nT > T_STOP;   // never for a finished pixel (nT < 0)
                const bool stop = ev.pass && !(nT > T_STOP);
                // (skipping the channel FMAs when no pixel of the quadrant passes -- as the backward pass does -- was
                // measured: +9 us per launch, the ballot + branch per evaluation costs more than the ~16 instructions
                // it saves on the few empty quadrants the reach masks let through)
                const float w = blend ? ev.alpha * T[k] : 0.f;
#pragma unroll
                for (int c = 0; c < CD; ++c) acc[k][c] = __fmaf_rn(rec[6 + c], w, acc[k][c]);
                T[k] = blend ? nT : (stop ? -fabsf(T[k]) : T[k]);
                last[k] = blend ? list_idx : last[k];
                if (__builtin_amdgcn_ballot_w64(T[k] > 0.f) == 0ull) alive &= ~(1u << k);
            }
            if (alive == 0u) break;
        }
    }
    unsigned inside_mask = 0u;
    float alpha_out[PPL];
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        alpha_out[k] = 0.f;
        if (!(pxi[k] < width && pyi[k] < height)) continue;
        inside_mask |= 1u << k;
        const float Tk = fabsf(T[k]);
        const size_t pix = ((size_t)cam * height + pyi[k]) * width + pxi[k];
        alphas[pix] = 1.f - Tk;
        alpha_out[k] = 1.f - Tk;
        last_ids[pix] = last[k];
        float* out = render + pix * CD;
#pragma unroll
        for (int c = 0; c < CD; ++c) {
            float v = acc[k][c];
            if (backgrounds) v = __fmaf_rn(Tk, backgrounds[cam * CD + c], v);
            out[c] = v;
            acc[k][c] = v;
        }
    }
    if constexpr (DECODE && CD == 10) {
        const RayCam rc = load_raycam(dec->intr + cam * dec->intr_stride, dec->c2w + cam * dec->c2w_stride);
        decode_epilogue<PPL, false>(*dec, rc, cam, pxi, pyi, inside_mask, width, height, acc, alpha_out);
    }
}

template <int CD, bool FILTER>
__global__ void __launch_bounds__(64 * TILES_PER_WG) __attribute__((amdgpu_waves_per_eu(CD <= 10 ? 5 : 1)))
raster_fwd_kernel(int n_tiles_total, int n_groups, int tile_w, int tile_h, int width, int height,
                  const float* __restrict__ records, const float* __restrict__ backgrounds,
                  const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
                  float* __restrict__ render, float* __restrict__ alphas, int32_t* __restrict__ last_ids,
                  const int32_t* __restrict__ tile_order, ClassSel cls, uint8_t* __restrict__ isect_reach) {
    constexpr int RQ = ((6 + CD + 3) & ~3) / 4;
    __shared__ float4 slab[TILES_PER_WG][64][RQ];
    __shared__ int idx_of[FILTER ? TILES_PER_WG : 1][64];
    __shared__ unsigned reach_of[TILES_PER_WG][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = scheduled_tile(tile_order, n_groups, n_tiles_total, wv);
    if (slot < 0) return;
    if (slot & SCHED_HEAVY)
        composite_fwd<CD, 1, FILTER>(slot & ~SCHED_HEAVY, wv, wv, lane, slab, idx_of, reach_of, cls, tile_w, tile_h, width, height,
                                     records, backgrounds, tile_offsets, flatten_ids, render, alphas, last_ids, isect_reach);
    else
        composite_fwd<CD, 4, FILTER>(slot, 0, wv, lane, slab, idx_of, reach_of, cls, tile_w, tile_h, width, height, records,
                                     backgrounds, tile_offsets, flatten_ids, render, alphas, last_ids, isect_reach);
}

// ---------------------------------------------------------------------------------------------------
// forward, block-walk formulation (round 3): sixteen independent 4x4-pixel workers per wave
// ---------------------------------------------------------------------------------------------------
// The kernel above evaluates a list entry at a whole 8x8 quadrant as soon as one of its pixels may pass the alpha
// test: on the benchmark lists 47 % of those lane-evaluations are live (profiles/r03/live_lane_stats.json), 69 % at
// 4x4-block granularity.  A wave cannot branch per 16 lanes, and gfx950 does not skip a 32-lane pass whose EXEC bits
// are all zero (scripts/ubench/exec_skip.hip: 1.93 cycles per v_fma whatever the mask) -- so finer culling needs lanes
// that work on DIFFERENT entries at the same time.  The forward pass has no cross-lane reduction, which makes that
// cheap here:
//   * worker j = the four lanes {j, j + 16, j + 32, j + 48}; it owns the 4x4 block (j & 3, j >> 2) of the tile, lane
//     a = lane >> 4 the four pixels of block row a (contiguous in x: 4 * CD contiguous floats per lane on the way out);
//   * a batch of 64 entries is staged as before (lane = entry); each lane also computes the 16-bit mask of blocks ITS
//     splat can reach (block_reach_mask16: per block a lower bound of sigma from its value and gradient at the block
//     centre -- convexity --, 10 VALU per block, lane-parallel over the 64 entries of the batch);
//   * sixteen ballots transpose the 64 x 16 bit matrix: worker j gets the 64-bit set of batch entries reaching its
//     block, and walks it with v_ffbl -- its own record read from the LDS slab each step (records padded to 80 bytes:
//     16 lanes of one ds_read_b128 group then hit distinct bank groups unless their entry indices agree mod 16);
//   * a lane leaves the walk when its four pixels are finished; the batch ends when every worker's set is empty.
// Every pixel still sees exactly the entries that can pass its alpha test, in list order, through the same
// instruction sequence: images are bit-identical to the quadrant kernel's.
// bit 4 * by + bx: the splat may reach alpha >= 1/255 at a pixel centre of the 4x4 block (bx, by) of tile (tx, ty).
// sigma is convex (the conic is positive definite), so over the block's pixel centres c + d, |d_x|, |d_y| <= 1.5:
//   sigma(c + d) >= sigma(c) + <grad sigma(c), d> >= sigma(c) - 1.5 (|g_x| + |g_y|),  g = (a dx + b dy, b dx + c dy).
// Conservative by construction plus the margin of reach_threshold(); tightened by the exact per-quadrant test.
__device__ inline unsigned block_reach_mask16(float mx, float my, float ca, float cb, float cc, float op, int tx,
                                              int ty) {
    const float thr = reach_threshold(op);
    if (thr < 0.f) return 0u;
    if (!(ca > 0.f && cc > 0.f && ca * cc - cb * cb > 0.f)) return 0xFFFFu;
    const float xc = (float)(tx * MOBGS_TILE) + 2.f, yc = (float)(ty * MOBGS_TILE) + 2.f;  // centre of block (0, 0)
    unsigned m = 0u;
#pragma unroll
    for (int by = 0; by < 4; ++by) {
        const float dy = my - (yc + (float)(4 * by));
#pragma unroll
        for (int bx = 0; bx < 4; ++bx) {
            const float dx = mx - (xc + (float)(4 * bx));
            const float gx = ca * dx + cb * dy, gy = cb * dx + cc * dy;
            const float lb = 0.5f * (dx * gx + dy * gy) - 1.5f * (fabsf(gx) + fabsf(gy));
            if (lb <= thr) m |= 1u << (4 * by + bx);
        }
    }
    const unsigned q = quadrant_reach_mask(mx, my, ca, cb, cc, op, tx, ty);
    return m & (((q & 1u) ? 0x0033u : 0u) | ((q & 2u) ? 0x00CCu : 0u) | ((q & 4u) ? 0x3300u : 0u) |
                ((q & 8u) ? 0xCC00u : 0u));
}
// ... from a staged record head (exponent form: common.h, write_splat_record)
__device__ inline unsigned block_reach_mask16_rec(const float4& r0, const float4& r1, int tx, int ty) {
    const ConicOp c = head_conic(r0, r1);
    return block_reach_mask16(r0.x, r0.y, c.ca, c.cb, c.cc, c.op, tx, ty);
}

#ifndef F2_WAVES
#define F2_WAVES 4
#endif
// FILTER: a class-restricted pass (ClassSel) -- entries of the other class simply get an empty block mask; list indices
// stay those of the combined list.
template <int CD, bool FILTER, bool DECODE = false>
__global__ void __launch_bounds__(64 * TILES_PER_WG) __attribute__((amdgpu_waves_per_eu(CD <= 10 ? F2_WAVES : (CD <= 12 ? 3 : 2))))
raster_fwd_blocks_kernel(int n_tiles_total, int n_groups, int tile_w, int tile_h, int width, int height,
                         const float* __restrict__ records, const float* __restrict__ backgrounds,
                         const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
                         float* __restrict__ render, float* __restrict__ alphas, int32_t* __restrict__ last_ids,
                         const int32_t* __restrict__ tile_order, ClassSel cls, uint8_t* __restrict__ isect_reach,
                         DecodeEpi dec) {
    const int all_reach = cls.all_reach;
    constexpr int RS = (6 + CD + 3) & ~3;
    constexpr int RQ = RS / 4;
    constexpr int RQP = RQ + 1;  // padded record: 16-byte quarters per slab row
    __shared__ float4 slab[TILES_PER_WG][64][RQP];
    __shared__ float4 hslab[TILES_PER_WG][64][RQ];   // heavy tiles keep the one-quadrant-per-wave walk
    __shared__ int hidx[FILTER ? TILES_PER_WG : 1][64];
    __shared__ unsigned hreach[TILES_PER_WG][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = scheduled_tile(tile_order, n_groups, n_tiles_total, wv);
    if (slot < 0) return;
    if (slot & SCHED_HEAVY) {
        composite_fwd<CD, 1, FILTER, DECODE>(slot & ~SCHED_HEAVY, wv, wv, lane, hslab, hidx, hreach, cls, tile_w, tile_h,
                                             width, height, records, backgrounds, tile_offsets, flatten_ids, render, alphas,
                                             last_ids, isect_reach, &dec);
        return;
    }
    const int tile = slot;
    const int tiles_per_cam = tile_w * tile_h;
    const int cam = tile / tiles_per_cam;
    const int tl = tile - cam * tiles_per_cam;
    const int ty = tl / tile_w, tx = tl - ty * tile_w;
    const int j = lane & 15, a = lane >> 4;
    const int pyi = ty * MOBGS_TILE + 4 * (j >> 2) + a;
    const int pxi0 = tx * MOBGS_TILE + 4 * (j & 3);
    const float py = (float)pyi + 0.5f;
    float px[4], T[4], acc[4][CD];
    int last[4];
    bool any_inside = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        px[k] = (float)(pxi0 + k) + 0.5f;
        const bool inside = pxi0 + k < width && pyi < height;
        any_inside = any_inside || inside;
        T[k] = inside ? 1.f : -1.f;  // T < 0: the pixel takes no more splats (see composite_fwd)
        last[k] = 0;
#pragma unroll
        for (int c = 0; c < CD; ++c) acc[k][c] = 0.f;
    }
    bool done = !any_inside;

    const int s = __builtin_amdgcn_readfirstlane(tile_offsets[tile]);
    const int e = __builtin_amdgcn_readfirstlane(tile_offsets[tile + 1]);

    constexpr int PQ = RQ < 2 ? RQ : 2;  // prefetched quarters of the next batch: position, conic, opacity
    float4 pre[PQ];
#pragma unroll
    for (int q = 0; q < PQ; ++q) pre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    int pre_g = -1;
    if (s + lane < e) {
        pre_g = flatten_ids[s + lane];
        const float4* r = reinterpret_cast<const float4*>(records + (size_t)pre_g * RS);
#pragma unroll
        for (int q = 0; q < PQ; ++q) pre[q] = r[q];
    }
    for (int b = s; b < e; b += 64) {
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
        wave_lds_fence();
        unsigned m16 = 0u;
        {
            float4 rest[RQ > PQ ? RQ - PQ : 1];
#pragma unroll
            for (int q = 0; q < (RQ > PQ ? RQ - PQ : 1); ++q) rest[q] = make_float4(0.f, 0.f, 0.f, 0.f);  // (keeps it in registers)
            if (RQ > PQ && pre_g >= 0) {
                const float4* r = reinterpret_cast<const float4*>(records + (size_t)pre_g * RS);
#pragma unroll
                for (int q = PQ; q < RQ; ++q) rest[q - PQ] = r[q];
            }
            if (pre_g >= 0) {
                m16 = all_reach ? 0xFFFFu
                                : block_reach_mask16_rec(pre[0], pre[1], tx, ty);
                const bool kept = !FILTER || cls.keeps(pre_g);
                if (!kept) m16 = 0u;
                // the backward pass walks quadrants: a quadrant is reachable iff one of its blocks is.  Class-restricted
                // passes of one render share ONE byte array (each pass owns the bytes of its class): never touch the
                // other class's entries
                if (isect_reach && kept) {
                    const unsigned q = ((m16 & 0x0033u) ? 1u : 0u) | ((m16 & 0x00CCu) ? 2u : 0u) |
                                       ((m16 & 0x3300u) ? 4u : 0u) | ((m16 & 0xCC00u) ? 8u : 0u);
                    isect_reach[b + lane] = (uint8_t)q;
                }
#pragma unroll
                for (int q = 0; q < PQ; ++q) slab[wv][lane][q] = pre[q];
#pragma unroll
                for (int q = PQ; q < RQ; ++q) slab[wv][lane][q] = rest[q - PQ];
            }
        }
        wave_lds_fence();
        pre_g = -1;
        if (b + 64 + lane < e) {
            pre_g = flatten_ids[b + 64 + lane];
            const float4* r = reinterpret_cast<const float4*>(records + (size_t)pre_g * RS);
#pragma unroll
            for (int q = 0; q < PQ; ++q) pre[q] = r[q];
        }
        // transpose: worker j's set of batch entries
        unsigned mlo = 0u, mhi = 0u;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64((m16 >> jj) & 1u);
            if (j == jj) {
                mlo = (unsigned)bal;
                mhi = (unsigned)(bal >> 32);
            }
        }
        if (done) mlo = mhi = 0u;
        while (true) {
            const bool has = (mlo | mhi) != 0u;
            if (__builtin_amdgcn_ballot_w64(has) == 0ull) break;
            if (has) {
                int i;
                if (mlo != 0u) {
                    i = __builtin_ctz(mlo);
                    mlo &= mlo - 1u;
                } else {
                    i = 32 + __builtin_ctz(mhi);
                    mhi &= mhi - 1u;
                }
                const int list_idx = b + i;
                float rec[RS];
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    const float4 v = slab[wv][i][q];
                    rec[4 * q] = v.x;
                    rec[4 * q + 1] = v.y;
                    rec[4 * q + 2] = v.z;
                    rec[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const Eval ev = eval_splat(rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], px[k], py);
                    const float nT = T[k] * (1.f - ev.alpha);
                    const bool blend = ev.pass && nT > T_STOP;   // never for a finished pixel (nT < 0)
                    const bool stop = ev.pass && !(nT > T_STOP);
                    const float w = blend ? ev.alpha * T[k] : 0.f;
#pragma unroll
                    for (int c = 0; c < CD; ++c) acc[k][c] = __fmaf_rn(rec[6 + c], w, acc[k][c]);
                    T[k] = blend ? nT : (stop ? -fabsf(T[k]) : T[k]);
                    last[k] = blend ? list_idx : last[k];
                }
                // all four pixels finished = all four sign bits set (T is never +0: it starts at +-1 and a live pixel
                // keeps T > 1e-4).  Three full-rate ANDs and one compare instead of the canonicalising v_max chain.
                if ((__float_as_int(T[0]) & __float_as_int(T[1]) & __float_as_int(T[2]) & __float_as_int(T[3])) < 0) {
                    done = true;
                    mlo = mhi = 0u;
                }
            }
        }
    }
    unsigned inside_mask = 0u;
    float alpha_out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        alpha_out[k] = 0.f;
        if (!(pxi0 + k < width && pyi < height)) continue;
        inside_mask |= 1u << k;
        const float Tk = fabsf(T[k]);
        const size_t pix = ((size_t)cam * height + pyi) * width + pxi0 + k;
        alphas[pix] = 1.f - Tk;
        alpha_out[k] = 1.f - Tk;
        last_ids[pix] = last[k];
        float* out = render + pix * CD;
#pragma unroll
        for (int c = 0; c < CD; ++c) {
            float v = acc[k][c];
            if (backgrounds) v = __fmaf_rn(Tk, backgrounds[cam * CD + c], v);
            out[c] = v;
            acc[k][c] = v;
        }
    }
    if constexpr (DECODE && CD == 10) {
        const RayCam rc = load_raycam(dec.intr + cam * dec.intr_stride, dec.c2w + cam * dec.c2w_stride);
        const int px4[4] = {pxi0, pxi0 + 1, pxi0 + 2, pxi0 + 3}, py4[4] = {pyi, pyi, pyi, pyi};
        decode_epilogue<4, true>(dec, rc, cam, px4, py4, inside_mask, width, height, acc, alpha_out);
    }
}

// ---------------------------------------------------------------------------------------------------
// backward, stage 1: per-(tile, splat) gradient records
// ---------------------------------------------------------------------------------------------------
// Wave-wide sum of NVP per-lane values with the cross-lane hardware of gfx950, no LDS traffic, no selects:
//   1. v_permlane32_swap pairs component i with i+NVP/2: one swap + one add leaves the sum over lane bit 5 of
//      the low-half components in lanes 0-31 and of the high-half components in lanes 32-63;
//   2. v_permlane16_swap does the same for lane bit 4 (odd/even rows of 16 lanes);
//   3. the NVP/4 survivors are all-reduced inside each 16-lane row with four DPP adds
//      (quad_perm xor 1, quad_perm xor 2, row_half_mirror, row_ror:8).
// Afterwards every lane of row r (= lane >> 4) holds, in v[0 .. NVP/4), the wave totals of components
//   (r >> 1) * NVP/2 + (r & 1) * NVP/4 + k.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row_allreduce(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm:[1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm:[2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x128>(v);  // row_ror:8
    return v;
}
template <int NVP>
__device__ __forceinline__ void wave_reduce_components(float (&v)[NVP]) {
    static_assert(NVP >= 8 && (NVP & (NVP - 1)) == 0, "NVP must be a power of two >= 8");
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
    for (int i = 0; i < NVP / 4; ++i) v[i] = row_allreduce(v[i]);
}

// NVP = 16: the same two swap stages, then HALVING steps inside each 16-lane row instead of four all-reduces of
// four registers: a DPP add written under a bank mask lets the two halves of a row keep different components, so
// 4 registers -> 2 (row_ror:8, lanes 8-15 take the partner register) -> 1 (row_half_mirror, banks 1/3 take the
// partner) in 6 instructions, and the last two quad_perm steps run on ONE register: 8 instead of 16 VALU.
// Afterwards every lane holds the wave total of component lane >> 2.
// (v_add_f32_dpp with a partial bank_mask keeps the destination in the masked-off lanes; the builtins cannot
// express that, hence the inline assembly; a DPP source written by the previous VALU needs 2 wait states and the
// compiler's hazard recogniser does not look inside the asm: the nops are placed by hand.)
//
// the two swap stages alone: afterwards lane l holds, in v[0..3], components 4 (l >> 4) + {0..3} summed over the four
// lanes {l & 15, (l & 15) + 16, + 32, + 48} -- the lanes of one block-walk worker
template <int N>
__device__ __forceinline__ void wave_reduce16_columns(float (&v)[N]) {
    static_assert(N >= 16, "reduces v[0..15]");
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %8\n\t"
        "v_permlane32_swap_b32 %1, %9\n\t"
        "v_permlane32_swap_b32 %2, %10\n\t"
        "v_permlane32_swap_b32 %3, %11\n\t"
        "v_permlane32_swap_b32 %4, %12\n\t"
        "v_permlane32_swap_b32 %5, %13\n\t"
        "v_permlane32_swap_b32 %6, %14\n\t"
        "v_permlane32_swap_b32 %7, %15\n\t"
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
          "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += v[i + 8];
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %4\n\t"
        "v_permlane16_swap_b32 %1, %5\n\t"
        "v_permlane16_swap_b32 %2, %6\n\t"
        "v_permlane16_swap_b32 %3, %7\n\t"
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += v[i + 4];
}
template <int N>
__device__ __forceinline__ float wave_reduce16_scatter(float (&v)[N]) {
    static_assert(N >= 16, "reduces v[0..15]");
    // The swaps exchange register halves IN PLACE; through the builtin the compiler copies one operand of every swap
    // first (7 v_mov + a 2-cycle bubble each), in assembly the sixteen sums are simply consumed where they lie.
    // The whole reduction is ONE block -- 8 + 4 swaps with their adds, then the halving steps inside the row with the
    // two independent chains (u0, u1) interleaved: a VALU result needs 2 wait states before a swap or a DPP source
    // reads it and the compiler's hazard recogniser does not look inside the asm, so the order keeps every consumer
    // at least three instructions behind its producer and nops stand only in front (the sums were written by the FMAs
    // just before) and inside the final dependent chain.  Against one statement with its own nops per step:
    // raster_bwd<10> 492 -> 484 us (scripts/ubench/swap_cost.hip: 195 -> 183 cycles per reduction for the tail alone).
    float u0, u1, w;
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %3, %11\n\t"
        "v_permlane32_swap_b32 %4, %12\n\t"
        "v_permlane32_swap_b32 %5, %13\n\t"
        "v_permlane32_swap_b32 %6, %14\n\t"
        "v_permlane32_swap_b32 %7, %15\n\t"
        "v_permlane32_swap_b32 %8, %16\n\t"
        "v_permlane32_swap_b32 %9, %17\n\t"
        "v_permlane32_swap_b32 %10, %18\n\t"
        "v_add_f32 %3, %3, %11\n\t"
        "v_add_f32 %7, %7, %15\n\t"
        "v_add_f32 %4, %4, %12\n\t"
        "v_add_f32 %8, %8, %16\n\t"
        "v_add_f32 %5, %5, %13\n\t"
        "v_add_f32 %9, %9, %17\n\t"
        "v_add_f32 %6, %6, %14\n\t"
        "v_add_f32 %10, %10, %18\n\t"
        "s_nop 0\n\t"
        "v_permlane16_swap_b32 %3, %7\n\t"
        "v_permlane16_swap_b32 %4, %8\n\t"
        "v_permlane16_swap_b32 %5, %9\n\t"
        "v_permlane16_swap_b32 %6, %10\n\t"
        "v_add_f32 %3, %3, %7\n\t"
        "v_add_f32 %4, %4, %8\n\t"
        "v_add_f32 %5, %5, %9\n\t"
        "v_add_f32 %6, %6, %10\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %2, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %2, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1"
        : "=&v"(u0), "=&v"(u1), "=&v"(w), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]),
          "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
    return w;
}

// One or two further components (NV = 17, 18: e.g. 9 features + 2 flow channels + depth) next to the sixteen above:
// one permlane32 swap + add puts the bit-5 sums of e0 into lanes 0-31 and of e1 into lanes 32-63, four DPP adds
// all-reduce every 16-lane row, one permlane16 swap of the register with itself + add folds the two rows of each
// half: 8 instructions, against 80 for the generic 32-component reduction such a kernel used before.  Afterwards
// lanes 0-31 hold the wave total of e0, lanes 32-63 that of e1.
__device__ __forceinline__ float wave_reduce_pair(float e0, float e1) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(e0), __float_as_uint(e1), false, false);
    float x = row_allreduce(__uint_as_float(r[0]) + __uint_as_float(r[1]));
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// One (pixel, splat) pair of the backward pass: rebuilds the transmittance in front of the splat, forms the
// cotangent of alpha and adds the pair's share of the 6 + CD gradient components to g (FIRST: g is written, not
// accumulated; unused -- selecting the variant per entry makes the compiler shuffle the 16 sums between two
// register sets, which costs more than the zero fill it saves).
// tvab = Tf * (v_alpha_out - <background, v_out>): both terms enter v_alpha as ra * Tf * (...)
// An evaluation for the quadrant backward that was BUILT, MEASURED AND NOT KEPT (round 6; -DMOBGS_BWD_CLAMPFREE_EVAL): raster_bwd<10>
// 442.4 / 443.1 us with the plain evaluation against 460.5 / 458.5 us with this one, alternating runs on one box
// (profiles/r06/ab_clampfree_eval.txt) -- the asm statement needs `alpha` and `raw` in two registers (a v_mov per
// quadrant that the plain form does not have), clobbers VCC in the middle of the compare chain and is a scheduling
// barrier between v_exp_f32 and its consumers: two half-rate instructions saved, more than that lost.  Kept as the A/B
// arm and because tests/test_gpu_clamp_flag.py (opacities around 0.999 against the C oracle) came with it.
// The idea: the clamp alpha = min(0.999, raw) and the test "the clamp is not active"
// (raw <= 0.999: else alpha has zero slope) can only matter for an entry whose OPACITY exceeds 0.999 -- raw = opacity *
// exp(-sigma) <= opacity wherever the pair passes (sigma >= 0) -- and the opacity is a property of the entry, wave-uniform.
// `maybe_clamped` (an SGPR: L = log2(opacity) >= CLAMP_L_MIN, decided when the entry is staged) sends such entries through
// three instructions behind a scalar branch; every other entry -- all but a handful of a scene -- skips v_min, v_cmp and the
// mask AND of every evaluated quadrant (2 half-rate VALU of ~51): alpha = raw and live = pass hold EXACTLY there, so every
// gradient is bit-identical.  -> e.raw is "opacity * visibility where alpha has a slope, else 0".
// CLAMP_L_MIN = log2(0.999) - 1e-3: far more than the 1 ulp of v_exp_f32 below log2(0.999).
constexpr float CLAMP_L_MIN = -0.0014434f - 1e-3f;
__device__ __forceinline__ Eval eval_splat_bwd(float gx, float gy, float A, float B, float C, float L, float px, float py,
                                               int maybe_clamped) {
    Eval e;
    e.dx = gx - px;
    e.dy = gy - py;
    float s = __fmaf_rn(A * e.dx, e.dx, L);
    s = __fmaf_rn(C * e.dy, e.dy, s);
    s = __fmaf_rn(B * e.dx, e.dy, s);
    e.raw = __builtin_amdgcn_exp2f(s);
    e.alpha = e.raw;
    asm volatile(
        "s_cmp_eq_u32 %[mc], 0\n\t"
        "s_cbranch_scc1 1f\n\t"
        "v_cmp_ge_f32 vcc, 0x3f7fbe77, %[raw]\n\t"       // 0.999 >= raw: the clamp is not active
        "v_min_f32 %[a], 0x3f7fbe77, %[a]\n\t"
        "v_cndmask_b32 %[raw], 0, %[raw], vcc\n"
        "1:"
        : [a] "+v"(e.alpha), [raw] "+v"(e.raw)
        : [mc] "s"(maybe_clamped)
        : "vcc", "scc");
    e.pass = !(s > L || e.alpha < ALPHA_MIN);
    return e;
}

// SPARSE (round 6; raster_shared.h dead_channels<CD>): `skip_dead` (wave-uniform, an SGPR) says that the entry is a static
// splat whose dead channels are exactly zero and whose gradient nobody wants.  Their two FMAs per channel and pair then sit
// behind ONE scalar branch inside a single asm statement -- the compiler sees one body with one register assignment
// (two C++ copies of the quadrant loop selected per entry were built first: the register allocator reconciled the two
// bodies' assignments with ~20 copies per quadrant exit and spilled 16 registers: raster_bwd<10> 456 -> 522 us).  The FMAs
// keep their place in the dot chain (c = 0..5, [6..], 9), so a dense entry computes exactly what it did, and for a
// skipped one fma(0, v, dot) = dot: every sum is unchanged bit for bit, the dead sums stay at their cleared zeros.
// RAW0: `ev` comes from eval_splat_bwd (ev.raw is already 0 where the clamp is active): live = pass.
template <int CD, int RS, int NVP, bool FIRST, bool SPARSE = false, bool RAW0 = false>
__device__ __forceinline__ void blend_bwd(const float (&rec)[RS], const Eval& ev, bool pass, float& T, float& behind,
                                          float tvab, const float (&vo)[CD], float (&g)[NVP], int skip_dead = 0) {
    auto acc = [](float& dst, float a, float b) { dst = FIRST ? a * b : __fmaf_rn(a, b, dst); };
    const float alpha = pass ? ev.alpha : 0.f;
    // 1 / (1 - alpha): the hardware reciprocal (1 ulp; 1 - alpha >= 1e-3).  A Newton step behind it (0.5 ulp) changes
    // no gradient at the third digit of its distance to the exact float64 value (scripts/exp_form_accuracy.py) -- the
    // product T * ra rounds once per step anyway -- and costs 9 us of the 460 (docs/COMPOSITOR_NOTES.md)
    const float om = 1.f - alpha;
    const float ra = __builtin_amdgcn_rcpf(om);
    T *= ra;
    const float fac = alpha * T;
    float dot = 0.f;
    constexpr int ND = SPARSE ? dead_channels<CD>() : 0;
    static_assert(!(SPARSE && FIRST), "the short body accumulates");
#pragma unroll
    for (int c = 0; c < CD; ++c) {
        if (ND > 0 && c >= DEAD_FIRST && c < DEAD_FIRST + ND) {
            if (c > DEAD_FIRST) continue;   // the whole dead group is emitted at its first channel
            constexpr int F = 6 + DEAD_FIRST;
            if constexpr (ND == 3) {
                asm volatile(
                    "s_cmp_lg_u32 %[sk], 0\n\t"
                    "s_cbranch_scc1 1f\n\t"
                    "v_fmac_f32 %[g0], %[fac], %[v0]\n\t"
                    "v_fmac_f32 %[d], %[c0], %[v0]\n\t"
                    "v_fmac_f32 %[g1], %[fac], %[v1]\n\t"
                    "v_fmac_f32 %[d], %[c1], %[v1]\n\t"
                    "v_fmac_f32 %[g2], %[fac], %[v2]\n\t"
                    "v_fmac_f32 %[d], %[c2], %[v2]\n"
                    "1:"
                    : [g0] "+v"(g[F]), [g1] "+v"(g[F + 1]), [g2] "+v"(g[F + 2]), [d] "+v"(dot)
                    : [sk] "s"(skip_dead), [fac] "v"(fac), [c0] "v"(rec[F]), [c1] "v"(rec[F + 1]), [c2] "v"(rec[F + 2]),
                      [v0] "v"(vo[DEAD_FIRST]), [v1] "v"(vo[DEAD_FIRST + 1]), [v2] "v"(vo[DEAD_FIRST + 2])
                    : "scc");
            } else if constexpr (ND == 5) {
                asm volatile(
                    "s_cmp_lg_u32 %[sk], 0\n\t"
                    "s_cbranch_scc1 1f\n\t"
                    "v_fmac_f32 %[g0], %[fac], %[v0]\n\t"
                    "v_fmac_f32 %[d], %[c0], %[v0]\n\t"
                    "v_fmac_f32 %[g1], %[fac], %[v1]\n\t"
                    "v_fmac_f32 %[d], %[c1], %[v1]\n\t"
                    "v_fmac_f32 %[g2], %[fac], %[v2]\n\t"
                    "v_fmac_f32 %[d], %[c2], %[v2]\n\t"
                    "v_fmac_f32 %[g3], %[fac], %[v3]\n\t"
                    "v_fmac_f32 %[d], %[c3], %[v3]\n\t"
                    "v_fmac_f32 %[g4], %[fac], %[v4]\n\t"
                    "v_fmac_f32 %[d], %[c4], %[v4]\n"
                    "1:"
                    : [g0] "+v"(g[F]), [g1] "+v"(g[F + 1]), [g2] "+v"(g[F + 2]), [g3] "+v"(g[F + 3]), [g4] "+v"(g[F + 4]),
                      [d] "+v"(dot)
                    : [sk] "s"(skip_dead), [fac] "v"(fac), [c0] "v"(rec[F]), [c1] "v"(rec[F + 1]), [c2] "v"(rec[F + 2]),
                      [c3] "v"(rec[F + 3]), [c4] "v"(rec[F + 4]), [v0] "v"(vo[DEAD_FIRST]), [v1] "v"(vo[DEAD_FIRST + 1]),
                      [v2] "v"(vo[DEAD_FIRST + 2]), [v3] "v"(vo[DEAD_FIRST + 3]), [v4] "v"(vo[DEAD_FIRST + 4])
                    : "scc");
            }
            continue;
        }
        acc(g[6 + c], fac, vo[c]);
        dot = __fmaf_rn(rec[6 + c], vo[c], dot);
    }
    const float v_alpha = __fmaf_rn(T, dot, ra * (tvab - behind));
    const float ov = ev.raw;   // opacity * visibility, before the clamp
    const bool live = RAW0 ? pass : (pass && ov <= ALPHA_MAX);  // the clamp at 0.999 has zero slope
    const float v_sigma = live ? -ov * v_alpha : 0.f;
    // Geometry terms as RAW sums: sum v_sigma dx, sum v_sigma dy, sum v_sigma dx^2, sum v_sigma dx dy, sum v_sigma dy^2.
    // The conic is a constant of the SPLAT, so v_xy = (ca A + cb B, cb A + cc B) and the factor 1/2 of the conic's
    // diagonal commute with the sums over pixels AND over tiles: stage 2 applies them once per splat
    // (finish_geometry) -- 7 VALU per pair here instead of 11.
    const float dx = ev.dx, dy = ev.dy;
    const float t = v_sigma * dx, u = v_sigma * dy;
    g[0] = FIRST ? t : g[0] + t;
    g[1] = FIRST ? u : g[1] + u;
    acc(g[2], t, dx);
    acc(g[3], t, dy);
    acc(g[4], u, dy);
    g[5] = FIRST ? v_sigma : g[5] + v_sigma;   // vis * v_alpha (where live) = -v_sigma / opacity: stage 2 divides once
    if (FIRST) {
#pragma unroll
        for (int i = 6 + CD; i < NVP; ++i) g[i] = 0.f;
    }
    behind = __fmaf_rn(fac, dot, behind);
}

// LDS of one workgroup of the backward kernel
template <int CD>
struct BwdShared {
    static constexpr int RS = (6 + CD + 3) & ~3;
    float4 slab[TILES_PER_WG][64][RS / 4];  // the batch's splat records, one copy per wave
    int slot_of[TILES_PER_WG][64];          // gradient slot of each batch entry
    int idx_of[TILES_PER_WG][64];           // class-filtered passes: list index of each staged entry
    unsigned reach_of[TILES_PER_WG][64];    // quadrants of the tile each staged entry can reach
    float part[TILES_PER_WG][64][RS];       // heavy tiles: per-wave (= per-quadrant) gradient records of the batch
    unsigned long long touched[TILES_PER_WG];
    int top[TILES_PER_WG];
    float zero16[16] __attribute__((aligned(16)));  // 64 bytes of zeros: the accumulators are cleared by reading them
};

// The BACKWARD half of the decoder fusion (round 6; DECB): the Sandwich decoder's backward pass as the PROLOGUE of the backward
// compositor.  A lane needs the cotangent of the composited image at its four pixels before it can walk the list; instead of
// reading what a decoder_bwd launch wrote (40 B per pixel, after that launch read 56 B and wrote 44 B per pixel: 47 us +
// a 4.7-us reduction at 1352x1014), the lane reads the decoder's own inputs -- the composited features the forward pass kept,
// the cotangents of the decoded colour and of the expected depth -- and evaluates decoder_shared.h sandwich_backward for each
// pixel: the same instruction sequence as decoder_bwd_kernel, so the compositing loop sees the same bits.  The weight and
// pose gradients are sums over pixels of outer products; ALL of them come out of one 16x16 accumulator block of
// v_mfma_f32_16x16x4_f32 per wave (exact fp32):
//     rows  (A) = [vh_0..5 | vd_0..2 | h_0..5 | 0]            vh: hidden cotangents, vd: cotangent of the un-normalised ray
//     cols  (B) = [x_0..5 | dir_0..2 | loc_x, loc_y, 1 | vy_0..2 | 0]
//   g_w1[j][c]   = C[j][c] (c < 6: features), C[j][c - 3] (c >= 9: direction), t_(c-6) * C[j][11] (c = 6..8: the ray origin is
//                  the camera position t, one constant per image);  g_w2[o][j] = C[9 + j][12 + o];
//   pose: d/dR[i][0..1], d/d(third column)[i] = C[6 + i][9..11];  d/dt[i] = sum_j w1[j][6 + i] C[j][11] (vx is linear in vh)
// -- no per-lane accumulators, no wave reductions.  One partial row of 102 sums per TILE (w_partial[tile][102], the layout of
// decoder_bwd_kernel's rows: decoder_wgrad_reduce_kernel sums them, image by image for the pose).  Tiles with empty lists
// run the prologue too: their pixels still carry decoder gradients.
struct DecodeBwd {
    const float *feat, *v_rgb, *v_depth;   // [C,H,W,10] composited image, [C,3,H,W], [C,H,W] | NULL
    const float *intr, *c2w, *w1, *w2;     // as DecodeEpi
    float* w_partial;                      // [C * tiles, 102]
    unsigned* ticket;                      // the reduction's ticket word (cleared here)
    int intr_stride, c2w_stride;
};
constexpr int DECB_NRED = 102;

// One wave walks `tile` back to front for NP pixels per lane.  NP = 4: the whole tile, one gradient record per
// (tile, splat) straight to grad_slots.  NP = 1 (HEAVY): the 4 waves of the workgroup share the tile, wave `quad`
// taking one 8x8 quadrant; they walk the same batches in step, leave their partial records in LDS and the
// workgroup sums the (up to 4) partials of every entry into its ONE slot -- still no floating-point atomics, and a
// fixed summation order (quadrant 0..3), so gradients stay bit-reproducible.
template <int CD, int NP, bool FILTER, bool DECB = false>
__device__ __forceinline__ void composite_bwd(int tile, int quad, int wv, int lane, BwdShared<CD>& sh, ClassSel cls,
                                              int tile_w,
                                              int tile_h, int width, int height, const float* __restrict__ records,
                                              const float* __restrict__ backgrounds,
                                              const int32_t* __restrict__ radii, const int32_t* __restrict__ cum_tiles,
                                              const int32_t* __restrict__ keep_scan,
                                              const int32_t* __restrict__ tile_offsets,
                                              const int32_t* __restrict__ flatten_ids,
                                              const float* __restrict__ render_alphas,
                                              const int32_t* __restrict__ last_ids,
                                              const float* __restrict__ v_render, const float* __restrict__ v_alphas,
                                              float* __restrict__ grad_slots,
                                              const uint8_t* __restrict__ isect_reach,
                                              int32_t* __restrict__ any_record, const DecodeBwd& db = DecodeBwd{}) {
    constexpr int RS = (6 + CD + 3) & ~3;
    constexpr int RQ = RS / 4;
    constexpr int NV = 6 + CD;
    static_assert(!DECB || (CD == 10 && !FILTER), "decoder prologue: the 9 + 1 channel pass");
    // per-lane gradient sums of one entry: 6 + CD components, padded to what the wave reduction handles -- a power of
    // two, or sixteen plus 2 / plus 8 (NVX extra components reduced on their own, see below)
    constexpr int NVX = (NV > 16 && NV <= 24) ? NV - 16 : 0;
    constexpr int NVP = NV <= 8 ? 8 : (NV <= 16 ? 16 : (NVX ? (NVX <= 2 ? 18 : 24) : (NV <= 32 ? 32 : 64)));
    constexpr int PPL = NP;
    constexpr bool HEAVY = NP == 1;
    constexpr bool HAS_SPARSE = dead_channels<CD>() > 0;
    auto& slab = sh.slab;
    auto& slot_of = sh.slot_of;
    const int tiles_per_cam = tile_w * tile_h;
    const int cam = tile / tiles_per_cam;
    const int tl = tile - cam * tiles_per_cam;
    const int ty = tl / tile_w, tx = tl - ty * tile_w;

    const int s = __builtin_amdgcn_readfirstlane(tile_offsets[tile]);
    const int e = __builtin_amdgcn_readfirstlane(tile_offsets[tile + 1]);
    if (!DECB && e <= s) return;

    // behind[k] = sum over the splats BEHIND the current one of fac * <colour, v_out>: upstream keeps the
    // per-channel sums buffer[c] and forms sum_c (colour_c T - buffer_c ra) v_out_c; distributing v_out gives
    // T <colour, v_out> - ra * behind, one scalar per pixel instead of D (fewer registers, D fewer FMAs per pair)
    float px[PPL], py[PPL], T[PPL], Tf[PPL], va[PPL], bgdot[PPL], behind[PPL], tvab[PPL];
    float vo[PPL][CD];
    int binf[PPL];
    int top = -1;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    f32x4 wacc = {0.f, 0.f, 0.f, 0.f};   // DECB: C[4 (lane >> 4) + i][lane & 15] of the weight-gradient block
    RayCam rc;
    if constexpr (DECB) rc = load_raycam(db.intr + (size_t)cam * db.intr_stride, db.c2w + (size_t)cam * db.c2w_stride);
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int qd = HEAVY ? quad : k;
        const int pxi = tx * MOBGS_TILE + 8 * (qd & 1) + (lane & 7);
        const int pyi = ty * MOBGS_TILE + 8 * (qd >> 1) + (lane >> 3);
        px[k] = (float)pxi + 0.5f;
        py[k] = (float)pyi + 0.5f;
        const bool inside = pxi < width && pyi < height;
        binf[k] = -1;  // pixels outside the image never become valid
        Tf[k] = 1.f;
        va[k] = 0.f;
        bgdot[k] = 0.f;
        behind[k] = 0.f;
#pragma unroll
        for (int c = 0; c < CD; ++c) vo[k][c] = 0.f;
        if constexpr (DECB) {
            // every load unconditional, from clamped addresses (pixels past the image edge are computed and then dropped):
            // the weights arrive by scalar loads, which a divergent branch would not skip anyway
            const int cxi = min(pxi, width - 1), cyi = min(pyi, height - 1);
            const size_t P = (size_t)width * height;
            const size_t p = (size_t)cyi * width + cxi;
            const size_t pix = (size_t)cam * P + p;
            float f[10], vr[3], x6[6], loc[2], inv_n;
            const float* fp = db.feat + pix * 10;
#pragma unroll
            for (int c = 0; c < 10; ++c) f[c] = fp[c];
#pragma unroll
            for (int o = 0; o < 3; ++o) vr[o] = db.v_rgb[((size_t)cam * 3 + o) * P + p];
            const float gdep = db.v_depth ? db.v_depth[pix] : 0.f;
            const float alpha_px = render_alphas[pix];
            pixel_ray_xy(rc, cxi, cyi, x6, loc, inv_n);
            float vf[10], v_a, hh[6], vh[6], vy[3], vdir[3], vorg[3];
            sandwich_backward(db.w1, db.w2, f, x6, vr, alpha_px, gdep, vf, v_a, hh, vh, vy, vdir, vorg);
            (void)vorg;
            // dir = d / |d|: v_d = (v_dir - dir <dir, v_dir>) / |d|
            const float dotp = __fmaf_rn(x6[3], vdir[0], __fmaf_rn(x6[4], vdir[1], x6[5] * vdir[2]));
            float arow[16], brow[16];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                arow[j] = vh[j];
                arow[9 + j] = hh[j];
                brow[j] = f[3 + j];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                arow[6 + i] = (vdir[i] - x6[3 + i] * dotp) * inv_n;
                brow[6 + i] = x6[3 + i];
                brow[12 + i] = vy[i];
            }
            arow[15] = 0.f;
            brow[9] = loc[0];
            brow[10] = loc[1];
            brow[11] = 1.f;
            brow[15] = 0.f;
            if (inside) {
                binf[k] = last_ids[pix];
                Tf[k] = 1.f - alpha_px;
                va[k] = (v_alphas ? v_alphas[pix] : 0.f) + v_a;
#pragma unroll
                for (int c = 0; c < CD; ++c) vo[k][c] = vf[c];
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) arow[c] = 0.f;   // (a zero row: the pixel adds nothing to any sum)
            }
            // [16 x 64 px] x [64 px x 16] on the matrix pipe: lane (m, kq) supplies A[m][kq], B[kq][m] of pixels 4 s + kq
            float4* sa = &sh.slab[wv][lane][0];
            float4* sb = reinterpret_cast<float4*>(&sh.part[wv][lane][0]);
            wave_lds_fence();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sa[q] = make_float4(arow[4 * q], arow[4 * q + 1], arow[4 * q + 2], arow[4 * q + 3]);
                sb[q] = make_float4(brow[4 * q], brow[4 * q + 1], brow[4 * q + 2], brow[4 * q + 3]);
            }
            wave_lds_fence();
            const float* sa_f = reinterpret_cast<const float*>(&sh.slab[wv][0][0]);
            const float* sb_f = &sh.part[wv][0][0];
#pragma unroll
            for (int s4 = 0; s4 < 16; ++s4) {
                const int pxl = 4 * s4 + (lane >> 4);
                wacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sa_f[pxl * 16 + (lane & 15)], sb_f[pxl * 16 + (lane & 15)], wacc,
                                                            0, 0, 0);
            }
        }
        if (!DECB && inside) {
            const size_t pix = ((size_t)cam * height + pyi) * width + pxi;
            binf[k] = last_ids[pix];
            Tf[k] = 1.f - render_alphas[pix];
            va[k] = v_alphas ? v_alphas[pix] : 0.f;
            const float* vr = v_render + pix * CD;
#pragma unroll
            for (int c = 0; c < CD; ++c) vo[k][c] = vr[c];
        }
        if (inside) {
            if (backgrounds) {
#pragma unroll
                for (int c = 0; c < CD; ++c) bgdot[k] = __fmaf_rn(backgrounds[cam * CD + c], vo[k][c], bgdot[k]);
            }
            // a pixel whose cotangents are all exactly zero (masked-out loss, loss term with weight 0) contributes
            // nothing to any gradient: treat it like a pixel that blended nothing, so that it neither extends the
            // walk (top) nor passes a test -- tiles without any cotangent cost only this prologue
            bool nz = va[k] != 0.f;
#pragma unroll
            for (int c = 0; c < CD; ++c) nz = nz || (vo[k][c] != 0.f);
            if (!nz) binf[k] = -1;
            top = max(top, binf[k]);
        }
        T[k] = Tf[k];
        tvab[k] = Tf[k] * (va[k] - bgdot[k]);
    }
    if constexpr (DECB) {
        // the tile's partial row of weight / pose gradient sums, from the 16 x 16 block (layout: DecodeBwd)
        float* cm = reinterpret_cast<float*>(&sh.slab[wv][0][0]);   // this wave's block, row-major
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) cm[(4 * (lane >> 4) + i) * 16 + (lane & 15)] = wacc[i];
        wave_lds_fence();
        if (HEAVY) __syncthreads();   // the four quadrant waves of a heavy tile: wave 0 adds the four blocks
        if (!HEAVY || wv == 0) {
            auto C_ = [&](int r, int c) -> float {
                if (!HEAVY) return cm[r * 16 + c];
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < TILES_PER_WG; ++w) t += reinterpret_cast<const float*>(&sh.slab[w][0][0])[r * 16 + c];
                return t;
            };
            float* row = db.w_partial + (size_t)tile * DECB_NRED;
            for (int en = lane; en < DECB_NRED; en += 64) {
                float v;
                if (en < 72) {
                    const int j = en / 12, c = en - 12 * j;
                    const float tc = c == 6 ? rc.c2w[3] : (c == 7 ? rc.c2w[7] : rc.c2w[11]);   // (no run-time index: registers)
                    v = c < 6 ? C_(j, c) : (c < 9 ? tc * C_(j, 11) : C_(j, c - 3));
                } else if (en < 90) {
                    const int o = (en - 72) / 6, j = (en - 72) - 6 * o;
                    v = C_(9 + j, 12 + o);
                } else {
                    const int i = (en - 90) >> 2, q = (en - 90) & 3;
                    if (q < 3) {
                        v = C_(6 + i, 9 + q);
                    } else {
                        v = 0.f;
                        for (int j = 0; j < 6; ++j) v = __fmaf_rn(db.w1[12 * j + 6 + i], C_(j, 11), v);
                    }
                }
                row[en] = v;
            }
        }
        if (e <= s) return;
    }
    // highest list index any pixel of pixel slot k (= one 8x8 quadrant) blended: entries behind it cannot contribute
    // to that quadrant and are not even evaluated there (on the benchmark lists 21 % of the entries lie behind every
    // quadrant's last blended entry -- scripts/live_lane_stats.py)
    int topk[PPL];
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        int t = binf[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t = max(t, __shfl_xor(t, off, 64));
        topk[k] = __builtin_amdgcn_readfirstlane(t);
    }
    // ... and of the whole tile
    top = topk[0];
#pragma unroll
    for (int k = 1; k < PPL; ++k) top = max(top, topk[k]);
    if (HEAVY) {  // all 4 quadrant waves walk the same batches
        if (lane == 0) sh.top[wv] = top;
        __syncthreads();
        top = max(max(sh.top[0], sh.top[1]), max(sh.top[2], sh.top[3]));
    }
    top = min(top, e - 1);
    // tells stage 2 that there may be something to sum (see mobgs_raster_bwd): once per wave with a non-empty walk --
    // when every cotangent of the pass is zero no pixel is valid, top stays below s and nobody sets it.  (Setting it
    // at the first record instead cost 16 us of the 540: a branch per list entry.)
    if (any_record && top >= s && lane == 0) *any_record = 1;
    // cover (MobgsTuning.cover_slots): nobody zero-filled the slots.  The entries behind every pixel's last blended one
    // are never staged: their slots get their zeros here (lane = entry; the slot index by the same chain as below)
    const bool cover = !FILTER && cls.cover != 0;
    if (cover && (!HEAVY || quad == 0)) {
        // (base >= s: `top` = -1 when no pixel of the tile is valid -- far below a list that starts at s; without the second
        // condition such a tile walked down to index 0: the whole backward pass 0.50 -> 0.67 ms with a third of the image
        // without cotangents, 0.43 -> 1.16 ms with two thirds (scripts/r06/dead_tiles_probe.py).
        // Of five ways to write the bound this one disturbs the allocator's choices in the main loop least: +3 us on the lean
        // step against +4 .. +7, profiles/r06/ab_cover_preloop_bound.txt)
        for (int base = e - 1; base > top && base >= s; base -= 64) {
            const int idx = base - lane;
            if (idx > top && idx >= s) {
                const int g = flatten_ids[idx];
                const float2 m = *reinterpret_cast<const float2*>(records + (size_t)g * RS);
                const TileRect tr = tile_rect(m.x, m.y, radii[g], tile_w, tile_h);
                const int slot = keep_index(keep_scan, cum_tiles[g] + (ty - tr.y0) * (tr.x1 - tr.x0) + (tx - tr.x0));
                float4* dst = reinterpret_cast<float4*>(grad_slots + (size_t)slot * RS);
#pragma unroll
                for (int q = 0; q < RQ; ++q) dst[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }

    for (int hi = top; hi >= s; hi -= 64) {
        int n = min(64, hi - s + 1);
        unsigned long long touched = 0ull;  // heavy: batch entries this wave produced a record for
        wave_lds_fence();
        {
            const int g = lane < n ? flatten_ids[hi - lane] : 0;
            const bool keep = lane < n && (!FILTER || cls.keeps(g));
            int pos = lane;
            if (FILTER) {  // stage only the wanted class, compacted (still back to front)
                const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
                n = __builtin_popcountll(km);
                pos = __builtin_popcountll(km & ((1ull << lane) - 1ull));
            }
            if (keep) {
                const float4* r = reinterpret_cast<const float4*>(records + (size_t)g * RS);
                const float4 r0 = r[0], r1 = r[1];
                slab[wv][pos][0] = r0;
                slab[wv][pos][1] = r1;
                // bit 4: a static row with its dead channels at zero -- the entry takes the short blend body
                bool dead_zero = HAS_SPARSE && cls.static_row(g);
#pragma unroll
                for (int q = 2; q < RQ; ++q) {
                    const float4 v = r[q];
                    slab[wv][pos][q] = v;
                    if constexpr (HAS_SPARSE) dead_zero = dead_zero && quarter_dead_zero<CD>(q, v);
                }
                // bit 5: the opacity may reach the 0.999 clamp (eval_splat_bwd)
                const unsigned sparse_bit = (dead_zero ? 16u : 0u) | (r1.y >= CLAMP_L_MIN ? 32u : 0u);
                // the forward pass left the masks of these very lists behind (isect_reach); else recompute
                sh.reach_of[wv][pos] = sparse_bit |
                                       (isect_reach ? (unsigned)isect_reach[hi - lane]
                                        : cls.all_reach
                                            ? 0xFu
                                            : quadrant_reach_mask_rec(r0, r1, tx, ty));
                const TileRect tr = tile_rect(r0.x, r0.y, radii[g], tile_w, tile_h);
                slot_of[wv][pos] = keep_index(keep_scan, cum_tiles[g] + (ty - tr.y0) * (tr.x1 - tr.x0) + (tx - tr.x0));
                if (FILTER) sh.idx_of[wv][pos] = hi - lane;
            }
        }
        wave_lds_fence();
        // per pixel slot k, the batch entries that can reach its 8x8 quadrant at all (wave-uniform bit masks):
        // ~40 % of the (entry, quadrant) pairs of a typical list are out of reach and are never evaluated
        unsigned long long reach[PPL];
        unsigned long long sparse = 0ull;   // batch entries that take the short blend body (wave-uniform)
        unsigned long long clampy;          // ... whose opacity may reach the clamp
        {
            const unsigned rm = lane < n ? sh.reach_of[wv][lane] : 0u;
            if constexpr (HAS_SPARSE) sparse = __builtin_amdgcn_ballot_w64((rm & 16u) != 0u);
            clampy = __builtin_amdgcn_ballot_w64((rm & 32u) != 0u);
            const int my_idx = FILTER ? (lane < n ? sh.idx_of[wv][lane] : 0x7fffffff) : hi - lane;
#pragma unroll
            for (int k = 0; k < PPL; ++k)
                reach[k] = __builtin_amdgcn_ballot_w64(((rm >> (HEAVY ? quad : k)) & 1u) != 0u && my_idx <= topk[k]);
        }
        unsigned long long rem = reach[0];
#pragma unroll
        for (int k = 1; k < PPL; ++k) rem |= reach[k];
        while (rem != 0ull) {
            const int j = __builtin_ctzll(rem);
            rem &= rem - 1ull;
            const int idx = FILTER ? sh.idx_of[wv][j] : hi - j;
            // The sixteen per-entry accumulators are cleared by four LDS broadcast reads of a zeroed 64-byte cell instead of
            // sixteen v_mov: this loop is bound by VALU issue, the LDS pipe is not (raster_bwd 542 -> 535 us).  Issued
            // ahead of the record reads -- LDS returns in order, so the compiler's own wait for the record covers them
            // -- and consumed after an explicit s_waitcnt.
            using f4 = __attribute__((ext_vector_type(4))) float;
            f4 z0, z1, z2, z3;
            // (the 18- / 24-accumulator builds run 3 waves per SIMD and measured 1.5 - 4 % slower with it)
            constexpr bool LDS_CLEAR = NVP == 16;
            if constexpr (LDS_CLEAR) {  // issued ahead of the record reads; LDS returns in order
                const unsigned za = (unsigned)(size_t)(&sh.zero16[0]);
                asm volatile(
                    "ds_read_b128 %0, %4\n\t"
                    "ds_read_b128 %1, %4 offset:16\n\t"
                    "ds_read_b128 %2, %4 offset:32\n\t"
                    "ds_read_b128 %3, %4 offset:48"
                    // EARLY-CLOBBER outputs: the four reads share the address register and return asynchronously -- with a
                    // plain "=v" the allocator may give an output tuple the address's register (it did, in the class-
                    // restricted instance <10, true>: `ds_read_b128 v[34:37], v34`), and once the first read has landed the
                    // later ones fetch from LDS address 0 + offset: record data instead of zeros, timing-dependent garbage
                    // in the gradient sums of train-mode renders on grids of > 1024 tiles (found in round 6 by running the
                    // fixtures' tests on the benchmark's kernel selection; tests/test_gpu_static_rows.py train cases,
                    // tests/test_gpu_render_parity.py "headline" arm)
                    : "=&v"(z0), "=&v"(z1), "=&v"(z2), "=&v"(z3)
                    : "v"(za)
                    : "memory");
            }
            float rec[RS];
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const float4 v = slab[wv][j][q];
                rec[4 * q] = v.x;
                rec[4 * q + 1] = v.y;
                rec[4 * q + 2] = v.z;
                rec[4 * q + 3] = v.w;
            }
            // the entry's gradient slot, wave-uniform: fetched with the record and kept in an SGPR, so that the store
            // behind the reduction needs no LDS round trip and no 64-bit vector address arithmetic
            int slot_u = 0;
            if constexpr (!HEAVY) slot_u = __builtin_amdgcn_readfirstlane(slot_of[wv][j]);
            float g[NVP];
            if constexpr (LDS_CLEAR) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    g[i] = z0[i];
                    g[4 + i] = z1[i];
                    g[8 + i] = z2[i];
                    g[12 + i] = z3[i];
                }
#pragma unroll
                for (int i = 16; i < NVP; ++i) g[i] = 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < NVP; ++i) g[i] = 0.f;
            }
            bool contributed = false;  // wave-uniform
            // a static entry with its dead channels at zero: their FMAs are jumped over (blend_bwd, SPARSE)
            const int skip_dead = HAS_SPARSE ? (int)((sparse >> j) & 1ull) : 0;
            const int maybe_clamped = (int)((clampy >> j) & 1ull);
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                // one 8x8 quadrant: skipped as a whole when the splat cannot reach it or none of its pixels blends
                // this splat (wave-uniform branches); otherwise every lane runs the same arithmetic with alpha = 0
                // standing in for "this pixel does not blend it" -- T, behind and the sums then stay exactly as
                // they were (x * 1, + 0)
                if (!((reach[k] >> j) & 1ull)) continue;
#ifdef MOBGS_BWD_CLAMPFREE_EVAL   // A/B arm, measured SLOWER and not the default (eval_splat_bwd): scripts/ab.sh build
                                  // clampfree raster.hip -DMOBGS_BWD_CLAMPFREE_EVAL
                Eval ev = eval_splat_bwd(rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], px[k], py[k], maybe_clamped);
                constexpr bool RAW0 = true;
#else
                Eval ev = eval_splat(rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], px[k], py[k]);
                constexpr bool RAW0 = false;
                (void)maybe_clamped;
#endif
                const bool pass = ev.pass && (idx <= binf[k]);
                if (__builtin_amdgcn_ballot_w64(pass) == 0ull) continue;
                blend_bwd<CD, RS, NVP, false, HAS_SPARSE, RAW0>(rec, ev, pass, T[k], behind[k], tvab[k], vo[k], g,
                                                                skip_dead);
                contributed = true;
            }
            if (!contributed) continue;
            touched |= 1ull << j;
            if constexpr (NVP == 16) {
                // every lane ends up with the total of component lane >> 2: one 64-byte store from 16 lanes
                const float w = wave_reduce16_scatter(g);
                if ((lane & 3) == 0 && (lane >> 2) < RS) {
                    float* dst = HEAVY ? &sh.part[wv][j][0] : grad_slots + (size_t)slot_u * RS;
                    dst[lane >> 2] = w;
                }
            } else if constexpr (NVX > 0) {
                float* dst = HEAVY ? &sh.part[wv][j][0] : grad_slots + (size_t)slot_u * RS;
                if constexpr (NVX <= 2) {
                    const float x = wave_reduce_pair(g[16], g[17]);
                    if ((lane & 31) == 0 && (lane >> 5) < NVX) dst[16 + (lane >> 5)] = x;
                } else {
                    float ext[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) ext[i] = g[16 + i];
                    wave_reduce_components<8>(ext);  // row r: components 16 + (r >> 1) * 4 + (r & 1) * 2 + {0, 1}
                    const int base = 16 + (lane >> 5) * 4 + ((lane >> 4) & 1) * 2;
                    if ((lane & 15) == 0 && base < RS) *reinterpret_cast<float2*>(dst + base) = make_float2(ext[0], ext[1]);
                }
                const float w = wave_reduce16_scatter(g);
                if ((lane & 3) == 0) dst[lane >> 2] = w;
            } else {
                wave_reduce_components<NVP>(g);
                // lane 0 of each 16-lane row stores its NVP/4 consecutive components
                constexpr int Q = NVP / 4;
                const int base = (lane >> 5) * (NVP / 2) + ((lane >> 4) & 1) * Q;
                if ((lane & 15) == 0 && base < RS) {
                    float* dst = HEAVY ? &sh.part[wv][j][base] : grad_slots + (size_t)slot_u * RS + base;
                    if constexpr (Q == 2) {
                        *reinterpret_cast<float2*>(dst) = make_float2(g[0], g[1]);
                    } else {
#pragma unroll
                        for (int q = 0; q < Q; q += 4)
                            if (base + q < RS)
                                *reinterpret_cast<float4*>(dst + q) = make_float4(g[q], g[q + 1], g[q + 2], g[q + 3]);
                    }
                }
            }
        }
        if (!HEAVY && cover && lane < n && !((touched >> lane) & 1ull)) {   // staged, but no pixel blended it: zeros
            float4* dst = reinterpret_cast<float4*>(grad_slots + (size_t)slot_of[wv][lane] * RS);
#pragma unroll
            for (int q = 0; q < RQ; ++q) dst[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (HEAVY) {
            if (lane == 0) sh.touched[wv] = touched;
            __syncthreads();
            // entry r of the batch, 16-byte part q: sum of the partial records of the quadrants that touched it
            for (int t = threadIdx.x; t < n * RQ; t += 64 * TILES_PER_WG) {
                const int r = t / RQ, q = t - r * RQ;
                float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
                bool any = false;
#pragma unroll
                for (int w = 0; w < TILES_PER_WG; ++w) {
                    if ((sh.touched[w] >> r) & 1ull) {
                        const float4 v = reinterpret_cast<const float4*>(sh.part[w][r])[q];
                        acc4.x += v.x;
                        acc4.y += v.y;
                        acc4.z += v.z;
                        acc4.w += v.w;
                        any = true;
                    }
                }
                if (any || cover) reinterpret_cast<float4*>(grad_slots + (size_t)slot_of[0][r] * RS)[q] = acc4;
            }
            __syncthreads();  // the partial records and slot table are free for the next batch
        }
    }
}

template <int CD, bool FILTER, bool DECB = false>
__global__ void __launch_bounds__(64 * TILES_PER_WG) __attribute__((amdgpu_waves_per_eu(CD <= 10 ? 4 : (CD <= 16 ? 3 : 1))))
raster_bwd_kernel(int n_tiles_total, int n_groups, int tile_w, int tile_h, int width, int height,
                  const float* __restrict__ records, const float* __restrict__ backgrounds,
                  const int32_t* __restrict__ radii, const int32_t* __restrict__ cum_tiles,
                  const int32_t* __restrict__ keep_scan, const int32_t* __restrict__ tile_offsets,
                  const int32_t* __restrict__ flatten_ids, const float* __restrict__ render_alphas,
                  const int32_t* __restrict__ last_ids,
                  const float* __restrict__ v_render, const float* __restrict__ v_alphas,
                  float* __restrict__ grad_slots, const int32_t* __restrict__ tile_order, ClassSel cls,
                  const uint8_t* __restrict__ isect_reach, int32_t* __restrict__ any_record, DecodeBwd db) {
    __shared__ BwdShared<CD> sh;
    if (cls.gated_off()) return;  // every cotangent of this pass is zero (uniform over the launch)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 16) sh.zero16[threadIdx.x] = 0.f;
    if constexpr (DECB) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *db.ticket = 0u;
    }
    __syncthreads();
    const int slot = scheduled_tile(tile_order, n_groups, n_tiles_total, wv);
    if (slot < 0) return;
    if (slot & SCHED_HEAVY)  // workgroup-uniform: all 4 slots of a heavy workgroup carry the flag
        composite_bwd<CD, 1, FILTER, DECB>(slot & ~SCHED_HEAVY, wv, wv, lane, sh, cls, tile_w, tile_h, width, height, records,
                                           backgrounds, radii, cum_tiles, keep_scan, tile_offsets, flatten_ids,
                                           render_alphas, last_ids, v_render, v_alphas, grad_slots, isect_reach, any_record,
                                           db);
    else
        composite_bwd<CD, 4, FILTER, DECB>(slot, 0, wv, lane, sh, cls, tile_w, tile_h, width, height, records, backgrounds,
                                           radii, cum_tiles, keep_scan, tile_offsets, flatten_ids, render_alphas, last_ids,
                                           v_render, v_alphas, grad_slots, isect_reach, any_record, db);
}


// ---------------------------------------------------------------------------------------------------
// backward, block-walk formulation (round 3): the sixteen 4x4-pixel workers of raster_fwd_blocks_kernel, back to front
// ---------------------------------------------------------------------------------------------------
// Same lane layout and per-worker entry sets as the forward block walk.  What the backward pass adds is the sum of the
// 6 + CD gradient components of an entry over all pixels of the tile -- with sixteen workers on sixteen different
// entries there is no wave-wide reduction to amortise:
//   * a lane accumulates its four pixels in registers (the first one WRITES the sixteen sums: no zero fill);
//   * the two swap stages of the wave reduction (v_permlane32_swap / v_permlane16_swap, 24 VALU) sum over the four lanes
//     of every worker at once and leave lane (r, j) with components 4 r .. 4 r + 3 of worker j's partial record -- one
//     float4 per lane, no DPP stage, no select;
//   * the partial records of the <= 16 blocks of a (tile, entry) pair meet in a per-wave LDS accumulator (64 entries x
//     64 bytes per batch): read - add - write of one float4 per lane.  Two workers can be at the same entry in the same
//     step (neighbouring blocks walk similar lists), and LDS float atomics are no way out (ds_add_f32: ~170 cycles per
//     wave instruction, scripts/ubench/lds_atomic.hip): colliding workers are serialised with an INTEGER claim -- every
//     pending worker ds_min_u32's its index into the entry's claim word, the smallest wins the round, adds, releases the
//     word; the others retry.  Fixed winner => fixed summation order => bit-reproducible gradients, as before;
//   * after a batch lane i flushes entry i's accumulated record to its gradient slot (if anything was added) and
//     clears it.
// A worker stops evaluating entries behind the last one any of ITS sixteen pixels blended.
template <int CD>
struct BwdBlocksShared {
    static constexpr int RS = (6 + CD + 3) & ~3;
    float4 slab[TILES_PER_WG][64][RS / 4 + 1];   // records of the batch, padded to spread the ds_read_b128 bank groups
    float4 acc[TILES_PER_WG][64][RS / 4];        // per-entry gradient records of the batch
    unsigned claim[TILES_PER_WG][64];
    int slot_of[TILES_PER_WG][64];
};

template <int CD>
__device__ __forceinline__ void composite_bwd_blocks(int tile, int wv, int lane, BwdBlocksShared<CD>& sh, int all_reach,
                                                     int tile_w, int tile_h, int width, int height,
                                                     const float* __restrict__ records,
                                                     const float* __restrict__ backgrounds,
                                                     const int32_t* __restrict__ radii,
                                                     const int32_t* __restrict__ cum_tiles,
                                                     const int32_t* __restrict__ keep_scan,
                                                     const int32_t* __restrict__ tile_offsets,
                                                     const int32_t* __restrict__ flatten_ids,
                                                     const float* __restrict__ render_alphas,
                                                     const int32_t* __restrict__ last_ids,
                                                     const float* __restrict__ v_render,
                                                     const float* __restrict__ v_alphas, float* __restrict__ grad_slots,
                                                     int32_t* __restrict__ any_record) {
    constexpr int RS = (6 + CD + 3) & ~3;
    constexpr int RQ = RS / 4;
    static_assert(RS == 16, "block-walk backward: 16-float records");
    const int tiles_per_cam = tile_w * tile_h;
    const int cam = tile / tiles_per_cam;
    const int tl = tile - cam * tiles_per_cam;
    const int ty = tl / tile_w, tx = tl - ty * tile_w;
    const int s = __builtin_amdgcn_readfirstlane(tile_offsets[tile]);
    const int e = __builtin_amdgcn_readfirstlane(tile_offsets[tile + 1]);
    if (e <= s) return;

    const int j = lane & 15, r = lane >> 4;
    const int pyi = ty * MOBGS_TILE + 4 * (j >> 2) + r;
    const int pxi0 = tx * MOBGS_TILE + 4 * (j & 3);
    const float py = (float)pyi + 0.5f;
    float px[4], T[4], behind[4], tvab[4], vo[4][CD];
    int binf[4];
    int top_w = -1;  // last list index any pixel of this lane blended
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        px[k] = (float)(pxi0 + k) + 0.5f;
        const bool inside = pxi0 + k < width && pyi < height;
        binf[k] = -1;
        float Tf = 1.f, va = 0.f, bgdot = 0.f;
        behind[k] = 0.f;
#pragma unroll
        for (int c = 0; c < CD; ++c) vo[k][c] = 0.f;
        if (inside) {
            const size_t pix = ((size_t)cam * height + pyi) * width + pxi0 + k;
            binf[k] = last_ids[pix];
            Tf = 1.f - render_alphas[pix];
            va = v_alphas ? v_alphas[pix] : 0.f;
            const float* vr = v_render + pix * CD;
#pragma unroll
            for (int c = 0; c < CD; ++c) vo[k][c] = vr[c];
            if (backgrounds) {
#pragma unroll
                for (int c = 0; c < CD; ++c) bgdot = __fmaf_rn(backgrounds[cam * CD + c], vo[k][c], bgdot);
            }
            bool nz = va != 0.f;  // all-zero cotangents: the pixel contributes to no gradient (see composite_bwd)
#pragma unroll
            for (int c = 0; c < CD; ++c) nz = nz || (vo[k][c] != 0.f);
            if (!nz) binf[k] = -1;
            top_w = max(top_w, binf[k]);
        }
        T[k] = Tf;
        tvab[k] = Tf * (va - bgdot);
    }
    // ... of this worker (lanes j, j + 16, j + 32, j + 48) ...
    top_w = max(top_w, __shfl_xor(top_w, 16, 64));
    top_w = max(top_w, __shfl_xor(top_w, 32, 64));
    // ... and of the tile
    int top = top_w;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) top = max(top, __shfl_xor(top, off, 64));
    top = min(__builtin_amdgcn_readfirstlane(top), e - 1);
    if (top < s) return;
    if (any_record && lane == 0) *any_record = 1;

    // clear this wave's accumulators and claim words
#pragma unroll
    for (int q = 0; q < RQ; ++q) sh.acc[wv][lane][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    sh.claim[wv][lane] = 0xFFFFFFFFu;

    for (int hi = top; hi >= s; hi -= 64) {
        const int n = min(64, hi - s + 1);
        wave_lds_fence();
        unsigned m16 = 0u;
        if (lane < n) {
            const int g = flatten_ids[hi - lane];
            const float4* rp = reinterpret_cast<const float4*>(records + (size_t)g * RS);
            const float4 r0 = rp[0], r1 = rp[1];
            sh.slab[wv][lane][0] = r0;
            sh.slab[wv][lane][1] = r1;
#pragma unroll
            for (int q = 2; q < RQ; ++q) sh.slab[wv][lane][q] = rp[q];
            m16 = all_reach ? 0xFFFFu : block_reach_mask16_rec(r0, r1, tx, ty);
            const TileRect tr = tile_rect(r0.x, r0.y, radii[g], tile_w, tile_h);
            sh.slot_of[wv][lane] = keep_index(keep_scan, cum_tiles[g] + (ty - tr.y0) * (tr.x1 - tr.x0) + (tx - tr.x0));
        }
        wave_lds_fence();
        // worker j's set of batch entries (bit i = entry hi - i: ascending bits walk back to front), cut at the last
        // entry one of its pixels blended
        unsigned mlo = 0u, mhi = 0u;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64((m16 >> jj) & 1u);
            if (j == jj) {
                mlo = (unsigned)bal;
                mhi = (unsigned)(bal >> 32);
            }
        }
        {
            const int cut = hi - top_w;  // entries i < cut lie behind the worker's last blended entry
            if (cut >= 64) {
                mlo = mhi = 0u;
            } else if (cut >= 32) {
                mlo = 0u;
                mhi &= ~0u << (cut - 32);
            } else if (cut > 0) {
                mlo &= ~0u << cut;
            }
        }
        while (true) {
            const bool has = (mlo | mhi) != 0u;
            if (__builtin_amdgcn_ballot_w64(has) == 0ull) break;
            if (has) {
                int i;
                if (mlo != 0u) {
                    i = __builtin_ctz(mlo);
                    mlo &= mlo - 1u;
                } else {
                    i = 32 + __builtin_ctz(mhi);
                    mhi &= mhi - 1u;
                }
                const int idx = hi - i;
                float rec[RS];
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    const float4 v = sh.slab[wv][i][q];
                    rec[4 * q] = v.x;
                    rec[4 * q + 1] = v.y;
                    rec[4 * q + 2] = v.z;
                    rec[4 * q + 3] = v.w;
                }
                float g[16];
                {
                    const Eval ev = eval_splat(rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], px[0], py);
                    blend_bwd<CD, RS, 16, true>(rec, ev, ev.pass && idx <= binf[0], T[0], behind[0], tvab[0], vo[0], g);
                }
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    const Eval ev = eval_splat(rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], px[k], py);
                    blend_bwd<CD, RS, 16, false>(rec, ev, ev.pass && idx <= binf[k], T[k], behind[k], tvab[k], vo[k], g);
                }
                // NOTE: the swaps below exchange registers between ALL lanes of the wave; lanes outside this branch
                // hold garbage in g, which only ever reaches the columns of their own (idle) workers
                wave_reduce16_columns(g);
                float4 part = make_float4(g[0], g[1], g[2], g[3]);
                bool pending = true;
                while (true) {
                    if (pending && r == 0) atomicMin(&sh.claim[wv][i], (unsigned)j);
                    const unsigned w = sh.claim[wv][i];
                    const bool go = pending && w == (unsigned)j;
                    if (go) {
                        float4 a4 = sh.acc[wv][i][r];
                        a4.x += part.x;
                        a4.y += part.y;
                        a4.z += part.z;
                        a4.w += part.w;
                        sh.acc[wv][i][r] = a4;
                        if (r == 0) sh.claim[wv][i] = 0xFFFFFFFFu;
                    }
                    pending = pending && !go;
                    if (__builtin_amdgcn_ballot_w64(pending) == 0ull) break;
                }
            }
        }
        // flush: lane i owns entry hi - i of the batch
        wave_lds_fence();
        if (lane < n) {
            float4 a4[RQ];
            bool any = false;
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                a4[q] = sh.acc[wv][lane][q];
                any = any || a4[q].x != 0.f || a4[q].y != 0.f || a4[q].z != 0.f || a4[q].w != 0.f;
            }
            if (any) {
                float4* dst = reinterpret_cast<float4*>(grad_slots + (size_t)sh.slot_of[wv][lane] * RS);
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    dst[q] = a4[q];
                    sh.acc[wv][lane][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    }
}

template <int CD>
__global__ void __launch_bounds__(64 * TILES_PER_WG) __attribute__((amdgpu_waves_per_eu(4)))
raster_bwd_blocks_kernel(int n_tiles_total, int n_groups, int tile_w, int tile_h, int width, int height,
                         const float* __restrict__ records, const float* __restrict__ backgrounds,
                         const int32_t* __restrict__ radii, const int32_t* __restrict__ cum_tiles,
                         const int32_t* __restrict__ keep_scan, const int32_t* __restrict__ tile_offsets,
                         const int32_t* __restrict__ flatten_ids, const float* __restrict__ render_alphas,
                         const int32_t* __restrict__ last_ids, const float* __restrict__ v_render,
                         const float* __restrict__ v_alphas, float* __restrict__ grad_slots,
                         const int32_t* __restrict__ tile_order, int all_reach,
                         const uint8_t* __restrict__ isect_reach, int32_t* __restrict__ any_record) {
    // heavy tiles (a whole workgroup per tile) keep the quadrant walk; the two never meet in one workgroup
    constexpr size_t LDS_BYTES = sizeof(BwdShared<CD>) > sizeof(BwdBlocksShared<CD>) ? sizeof(BwdShared<CD>)
                                                                                     : sizeof(BwdBlocksShared<CD>);
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = scheduled_tile(tile_order, n_groups, n_tiles_total, wv);
    const bool heavy = slot >= 0 && (slot & SCHED_HEAVY);  // workgroup-uniform: all 4 slots carry the flag
    if (heavy) {
        auto& sh = *reinterpret_cast<BwdShared<CD>*>(lds);
        if (threadIdx.x < 16) sh.zero16[threadIdx.x] = 0.f;
        __syncthreads();
        composite_bwd<CD, 1, false>(slot & ~SCHED_HEAVY, wv, wv, lane, sh, ClassSel{0, 1, 0, all_reach}, tile_w, tile_h,
                                    width, height, records, backgrounds, radii, cum_tiles, keep_scan, tile_offsets,
                                    flatten_ids, render_alphas, last_ids, v_render, v_alphas, grad_slots, isect_reach,
                                    any_record);
        return;
    }
    if (slot < 0) return;
    composite_bwd_blocks<CD>(slot, wv, lane, *reinterpret_cast<BwdBlocksShared<CD>*>(lds), all_reach, tile_w, tile_h,
                             width, height, records, backgrounds, radii, cum_tiles, keep_scan, tile_offsets,
                             flatten_ids, render_alphas, last_ids, v_render, v_alphas, grad_slots, any_record);
}


// ---------------------------------------------------------------------------------------------------
// backward, stage 0 (MobgsTuning.gate_zero_cotangent): is any cotangent of this pass non-zero?
// ---------------------------------------------------------------------------------------------------
// live[0] was cleared by the launcher; any thread that meets a non-zero element (NaN included: x != 0 holds) stores 1.
// The word is re-read once per grid-stride step, so with ordinary cotangents the kernel is over after the first wave
// of workgroups; with all-zero ones it streams both arrays once (12 channels x 8 cameras at 1352x1014: 0.53 GB).
__global__ void __launch_bounds__(256)
cotangent_probe_kernel(const float* __restrict__ a, size_t na, const float* __restrict__ b, size_t nb,
                       int32_t* __restrict__ live) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    bool nz = false;
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
        const float* p = which ? b : a;
        const size_t n = which ? nb : na;
        if (!p || n == 0) continue;
        const size_t head = min(n, (size_t)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4);
        const float4* p4 = reinterpret_cast<const float4*>(p + head);
        const size_t n4 = (n - head) / 4;
#pragma unroll 1
        for (size_t i = tid; i < n4; i += step) {
            if (__builtin_nontemporal_load(live) != 0) return;
            const float4 v = p4[i];
            nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
            if (nz) break;
        }
        if (tid < head) nz |= p[tid] != 0.f;
        const size_t tail = head + 4 * n4;
        if (tail + tid < n) nz |= p[tail + tid] != 0.f;
    }
    if (nz) *live = 1;
}
// the same question for a LIST of arrays (the cotangents of all outputs of a group of get_flow() calls): the table rides in
// the kernel arguments; blockIdx.y picks the array
constexpr int PROBE_MAX = 40;
struct ProbeTable {
    const float* p[PROBE_MAX];
    unsigned long long n[PROBE_MAX];
};
__global__ void __launch_bounds__(256) cotangent_probe_many_kernel(ProbeTable t, int32_t* __restrict__ live) {
    const float* p = t.p[blockIdx.y];
    const size_t n = t.n[blockIdx.y];
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    const size_t head = min(n, (size_t)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4);
    const float4* p4 = reinterpret_cast<const float4*>(p + head);
    const size_t n4 = (n - head) / 4;
    bool nz = false;
#pragma unroll 1
    for (size_t i = tid; i < n4; i += step) {
        if (__builtin_nontemporal_load(live) != 0) return;
        const float4 v = p4[i];
        nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
        if (nz) break;
    }
    if (tid < head) nz |= p[tid] != 0.f;
    const size_t tail = head + 4 * n4;
    if (tail + tid < n) nz |= p[tail + tid] != 0.f;
    if (nz) *live = 1;
}
// -> the gate word (any_record + 1), or nullptr when gating is off.  ONE extra launch: the caller's zero fill of the slot
// rows has cleared the two flag words behind them as well.  (A first version cleared the rows itself, and only when the
// pass was going to run -- memset + probe + gated clear: three launches per node, measured at +0.7 ms of HOST time per
// training iteration at 512x288, where the step is host-bound; the fill it saved matters only when the cotangents are
// zero, and then the host-side head of gaussian_renderer has dropped the node altogether.)
static const int32_t* arm_cotangent_gate(const MobgsTuning* tuning, const float* v_render, size_t n_render,
                                         const float* v_alphas, size_t n_alphas, int32_t* any_record, hipStream_t st) {
    if (!tuning_gate_zero_cotangent(tuning) || !any_record) return nullptr;
    int32_t* live = any_record + 1;
    const size_t work = (n_render + n_alphas) / 4 + 1;
    const int grid = (int)std::min<size_t>((work + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(cotangent_probe_kernel, dim3(grid), dim3(256), 0, st, v_render, n_render, v_alphas,
                       v_alphas ? n_alphas : 0, live);
    return live;
}

// ---------------------------------------------------------------------------------------------------
// backward, stage 2: per-splat sum of its slots -> dense gradient tensors
// ---------------------------------------------------------------------------------------------------
// Components 0..4 of a slot are raw sums (blend_bwd): A = sum v_sigma dx, B = sum v_sigma dy, Sxx = sum v_sigma dx^2,
// Sxy = sum v_sigma dx dy, Syy = sum v_sigma dy^2.  With the splat's conic (a, b, c) from its packed record:
//     v_mean2d = (a A + b B, b A + c B),   v_conic = (Sxx / 2, Sxy, Syy / 2);
// component 5 is S = sum v_sigma and v_opacity = -S / opacity = -S exp2(-L) (0 for a splat without slots or with S = 0:
// a pair only contributes when opacity * vis >= 1/255, so opacity > 0 wherever S != 0).  The record holds conic and
// opacity in exponent form (common.h, write_splat_record): record_conic_form() converts back.
// c0, c1, c2 = the summed components 0, 1, 2 in one lane -> v_x, v_y, v_conic_a (component 4 is halved by its lane).
// one past the last bounding-box intersection of splat g: cum_tiles[g + 1] when the intersections are enumerated in splat
// order; with a caller-chosen enumeration order (mobgs_project_and_bin_fused, enum_order) only "start + count" is right
__device__ __forceinline__ int box_end(const int32_t* __restrict__ cum_tiles, const int32_t* __restrict__ tiles_per_gauss,
                                       int g) {
    return tiles_per_gauss ? cum_tiles[g] + tiles_per_gauss[g] : cum_tiles[g + 1];
}
__device__ __forceinline__ void finish_geometry(const float* __restrict__ rec, bool any, float& c0, float& c1, float& c2) {
    if (any) {  // (a splat without slots may have no record at all: culled splats are never packed)
        float ca, cb, cc, op;
        record_conic_form(rec[2], rec[3], rec[4], rec[5], ca, cb, cc, op);  // the record holds the exponent form
        const float A = c0, B = c1;
        c0 = ca * A + cb * B;
        c1 = cb * A + cc * B;
    }
    c2 *= 0.5f;
}
template <int LPG>  // lanes per splat, >= record stride
__global__ void __launch_bounds__(256)
slot_reduce_kernel(int n_gauss, int channels, int has_extra, int stride, const int32_t* __restrict__ cum_tiles,
                   const int32_t* __restrict__ keep_scan, const float* __restrict__ grad_slots, float* __restrict__ v_means2d,
                   float* __restrict__ v_conics, float* __restrict__ v_opacities, float* __restrict__ v_colors,
                   float* __restrict__ v_extra, const int32_t* __restrict__ any_record,
                   const float* __restrict__ records, const int32_t* __restrict__ tiles_per_gauss) {
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / LPG;
    const int comp = threadIdx.x % LPG;
    if (gid >= n_gauss) return;
    // stage 1 wrote no record at all (every cotangent of the pass was zero): all sums are zero, read nothing
    const bool none = any_record && *any_record == 0;
    const int a = none ? 0 : keep_index(keep_scan, cum_tiles[gid]);
    const int b = none ? 0 : keep_index(keep_scan, box_end(cum_tiles, tiles_per_gauss, gid));
    float acc = 0.f;
    if (comp < stride) {
        // 4 independent partial sums keep 4 loads in flight per lane (the loop is latency-bound otherwise);
        // fixed association order -> still deterministic
        const float* p = grad_slots + (size_t)a * stride + comp;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = a;
        for (; k + 4 <= b; k += 4, p += 4 * stride) {
            s0 += p[0];
            s1 += p[stride];
            s2 += p[2 * stride];
            s3 += p[3 * stride];
        }
        for (; k < b; ++k, p += stride) s0 += *p;
        acc = (s0 + s1) + (s2 + s3);
    }
    const size_t g = (size_t)gid;
    {   // one lane per component: components 0 and 1 need each other (both lanes of the group are active here)
        const float other = __shfl_xor(acc, 1, 64);
        if (comp < 2) {
            if (b > a) {  // (no slots: the record may never have been packed)
                const float* rec = records + g * stride;
                float ca, cb, cc, op;
                record_conic_form(rec[2], rec[3], rec[4], rec[5], ca, cb, cc, op);
                acc = comp == 0 ? ca * acc + cb * other : cb * other + cc * acc;
            }
        } else if (comp == 2 || comp == 4) {
            acc *= 0.5f;
        }
        else if (comp == 5) acc = (b > a && acc != 0.f) ? -acc * __builtin_amdgcn_exp2f(-records[g * stride + 5]) : 0.f;
    }
    if (comp < 2)
        v_means2d[2 * g + comp] = acc;
    else if (comp < 5)
        v_conics[3 * g + (comp - 2)] = acc;
    else if (comp == 5)
        v_opacities[g] = acc;
    else if (comp - 6 < channels)
        v_colors[g * channels + (comp - 6)] = acc;
    else if (has_extra && comp - 6 == channels)
        v_extra[g] = acc;
}

// Records read as 16-byte quarters (strides 8, 12, 20 .. 32 floats: the 1-, 2-, 12- and 16-channel passes of get_flow):
// LPS lanes per splat, lane q < stride / 4 sums quarter q of every slot of the splat, four slots in flight per lane -- a
// quarter of the waves and of the load instructions of the one-float-per-lane kernel above (stride 20: 74 -> 40 us,
// stride 8: 25 -> 20 us at 300 k splats).  Fixed association order -> deterministic.
template <int LPS>
__global__ void __launch_bounds__(256)
slot_reduce_wide_kernel(int n_gauss, int channels, int has_extra, int rq, const int32_t* __restrict__ cum_tiles,
                        const int32_t* __restrict__ keep_scan, const float* __restrict__ grad_slots,
                        float* __restrict__ v_means2d, float* __restrict__ v_conics, float* __restrict__ v_opacities,
                        float* __restrict__ v_colors, float* __restrict__ v_extra,
                        const int32_t* __restrict__ any_record, const float* __restrict__ records,
                        const int32_t* __restrict__ tiles_per_gauss) {
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / LPS;
    const int q = threadIdx.x % LPS;
    if (gid >= n_gauss || q >= rq) return;
    const bool none = any_record && *any_record == 0;  // stage 1 wrote no record: all sums are zero
    const int a = none ? 0 : keep_index(keep_scan, cum_tiles[gid]);
    const int b = none ? 0 : keep_index(keep_scan, box_end(cum_tiles, tiles_per_gauss, gid));
    const float4* p = reinterpret_cast<const float4*>(grad_slots) + q;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    int k = a;
    for (; k + 4 <= b; k += 4) {
        const float4 u0 = p[(size_t)k * rq], u1 = p[(size_t)(k + 1) * rq], u2 = p[(size_t)(k + 2) * rq],
                     u3 = p[(size_t)(k + 3) * rq];
        s0.x += u0.x; s0.y += u0.y; s0.z += u0.z; s0.w += u0.w;
        s1.x += u1.x; s1.y += u1.y; s1.z += u1.z; s1.w += u1.w;
        s2.x += u2.x; s2.y += u2.y; s2.z += u2.z; s2.w += u2.w;
        s3.x += u3.x; s3.y += u3.y; s3.z += u3.z; s3.w += u3.w;
    }
    for (; k < b; ++k) {
        const float4 u = p[(size_t)k * rq];
        s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
    }
    float acc[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                    (s0.w + s1.w) + (s2.w + s3.w)};
    const size_t g = (size_t)gid;
    if (q == 0)       // components 0..3 live in quarter 0
        finish_geometry(records + g * 4 * rq, b > a, acc[0], acc[1], acc[2]);
    else if (q == 1) {  // component 4: the conic's c
        acc[0] *= 0.5f;
        acc[1] = (b > a && acc[1] != 0.f) ? -acc[1] * __builtin_amdgcn_exp2f(-records[g * 4 * rq + 5]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int comp = 4 * q + i;
        if (comp < 2)
            v_means2d[2 * g + comp] = acc[i];
        else if (comp < 5)
            v_conics[3 * g + (comp - 2)] = acc[i];
        else if (comp == 5)
            v_opacities[g] = acc[i];
        else if (comp - 6 < channels)
            v_colors[g * channels + (comp - 6)] = acc[i];
        else if (has_extra && comp - 6 == channels)
            v_extra[g] = acc[i];
    }
}

// 64-byte records (stride 16, the render() configuration): LPS lanes per splat, every lane loads 16 BYTES -- quarter
// q = lane & 3 of slot k + sub-group -- so one load instruction of a group covers LPS / 4 slots (the kernel is
// latency-bound: 4x the bytes in flight of a one-float-per-lane loop), two such loads in flight per lane; the
// partial sums per component are combined with DPP adds.  Measured: 16 lanes 33 us, 8 lanes (twice the splats per
// wave, half the waves to schedule) 30 us, 4 lanes 34 us.
// Fixed association order -> deterministic.
// WRED (round 6): the launch carries wgrad_finish_outputs(wf) * n_images extra LEADING workgroups that sum the decoder's
// weight / pose gradient rows the backward compositor's prologue left (decoder_shared.h wgrad_column_sum) -- a strided,
// latency-bound column each, finished long before the slot streams are: the 8-us reduction launch behind raster_bwd is gone.
template <int LPS, bool WRED = false>  // lanes per splat: 16 (four slots per load instruction and splat) or 8 (two)
__global__ void __launch_bounds__(256)
slot_reduce16_kernel(int n_gauss, int channels, int has_extra, const int32_t* __restrict__ cum_tiles,
                     const int32_t* __restrict__ keep_scan, const float* __restrict__ grad_slots,
                     float* __restrict__ v_means2d, float* __restrict__ v_conics, float* __restrict__ v_opacities,
                     float* __restrict__ v_colors, float* __restrict__ v_extra,
                     const int32_t* __restrict__ any_record, const float* __restrict__ records,
                     const int32_t* __restrict__ tiles_per_gauss, WgradFinish wf = WgradFinish{}) {
    // (With an enumeration order -- mobgs_hip.h, enum_order -- a splat's slots sit where the ORDER put them: this kernel
    // then reads 360-byte runs at random places instead of one stream, 32 -> 39 us.  Walking the splats in enumeration
    // order instead was measured: the slot buffer streams again, but cum_tiles / records are gathered and five small
    // outputs per splat scattered -- 65 us.)
    constexpr int SUBS = LPS / 4;
    int bid = blockIdx.x;
    if constexpr (WRED) {
        const int nout = wgrad_finish_outputs(wf), nw = nout * wf.n_images;
        if (bid < nw) {   // (workgroup-uniform)
            wgrad_column_sum(wf, bid % nout, bid / nout);
            return;
        }
        bid -= nw;
    }
    const int gid = (bid * blockDim.x + threadIdx.x) / LPS;
    const int l16 = threadIdx.x & (LPS - 1);
    const int q = l16 & 3, sub = l16 >> 2;
    const bool live = gid < n_gauss;
    int a = 0, b = 0;
    if (live && !(any_record && *any_record == 0)) {  // (no record written by stage 1: every sum is zero)
        a = keep_index(keep_scan, cum_tiles[gid]);
        b = keep_index(keep_scan, box_end(cum_tiles, tiles_per_gauss, gid));
    }
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    const float4* p = reinterpret_cast<const float4*>(grad_slots) + q;
    int k = a + sub;
    for (; k + SUBS < b; k += 2 * SUBS) {
        const float4 u = p[(size_t)k * 4], v = p[(size_t)(k + SUBS) * 4];
        s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
        s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
    }
    if (k < b) {
        const float4 u = p[(size_t)k * 4];
        s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
    }
    float acc[4] = {s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // sum over the sub-groups (lanes l, l^4[, l^8, l^12] of the row)
        if (SUBS == 4) {
            acc[i] = dpp_add<0x124>(acc[i]);  // row_ror:4
            acc[i] = dpp_add<0x128>(acc[i]);  // row_ror:8
        } else if (SUBS == 2) {
            // lane l of an 8-lane group with l ^ 4: row_half_mirror pairs l with 7 - l (other quarter!), so use the
            // two masked shifts instead: banks {0, 2} take lane + 4, banks {1, 3} lane - 4
            const float up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[i]), 0x104, 0xF, 0x5, true));
            const float dn = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[i]), 0x114, 0xF, 0xA, true));
            acc[i] = acc[i] + (up + dn);
        }
    }
    if (!live || sub != 0) return;
    const size_t g = (size_t)gid;
    if (q == 0)
        finish_geometry(records + g * 16, b > a, acc[0], acc[1], acc[2]);
    else if (q == 1) {
        acc[0] *= 0.5f;
        acc[1] = (b > a && acc[1] != 0.f) ? -acc[1] * __builtin_amdgcn_exp2f(-records[g * 16 + 5]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int comp = 4 * q + i;
        const float v = acc[i];
        if (comp < 2)
            v_means2d[2 * g + comp] = v;
        else if (comp < 5)
            v_conics[3 * g + (comp - 2)] = v;
        else if (comp == 5)
            v_opacities[g] = v;
        else if (comp - 6 < channels)
            v_colors[g * channels + (comp - 6)] = v;
        else if (has_extra && comp - 6 == channels)
            v_extra[g] = v;
    }
}

// supported total channel counts (compile-time accumulators); other counts are zero-padded by the host wrapper
template <typename F>
inline int dispatch_channels(int D, F&& f) {
    switch (D) {
        case 1: f(std::integral_constant<int, 1>{}); return MOBGS_OK;
        case 2: f(std::integral_constant<int, 2>{}); return MOBGS_OK;
        case 3: f(std::integral_constant<int, 3>{}); return MOBGS_OK;
        case 4: f(std::integral_constant<int, 4>{}); return MOBGS_OK;
        case 9: f(std::integral_constant<int, 9>{}); return MOBGS_OK;
        case 10: f(std::integral_constant<int, 10>{}); return MOBGS_OK;
        case 12: f(std::integral_constant<int, 12>{}); return MOBGS_OK;
        case 16: f(std::integral_constant<int, 16>{}); return MOBGS_OK;
        case 26: f(std::integral_constant<int, 26>{}); return MOBGS_OK;
        default: return MOBGS_E_UNSUPPORTED;
    }
}

}  // namespace mobgs

using namespace mobgs;

// MobgsTuning.quadrant_culling = 0 (testing aid): the compositors evaluate every quadrant of every entry (the
// per-quadrant reach masks only skip work that is predicated off at every pixel, so results must not change)

extern "C" {

int mobgs_cotangent_probe(int n_arrays, const float* const* arrays, const size_t* counts, int32_t* live, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n_arrays < 0 || !live || (n_arrays > 0 && (!arrays || !counts))) {
        set_error("mobgs_cotangent_probe: bad arguments");
        return MOBGS_E_INVALID;
    }
    (void)hipMemsetAsync(live, 0, sizeof(int32_t), st);
    for (int a0 = 0; a0 < n_arrays; a0 += PROBE_MAX) {
        ProbeTable t;
        int m = 0;
        size_t longest = 0;
        for (int a = a0; a < n_arrays && a < a0 + PROBE_MAX; ++a) {
            if (!arrays[a] || counts[a] == 0) continue;
            t.p[m] = arrays[a];
            t.n[m] = counts[a];
            longest = std::max(longest, counts[a]);
            ++m;
        }
        if (m == 0) continue;
        const int gx = (int)std::min<size_t>((longest / 4 + 256) / 256, 256);
        hipLaunchKernelGGL(cotangent_probe_many_kernel, dim3(gx, m), dim3(256), 0, st, t, live);
    }
    return check_launch("cotangent_probe_many_kernel");
}

int mobgs_raster_channels_supported(int D) {
    return D == 1 || D == 2 || D == 3 || D == 4 || D == 9 || D == 10 || D == 12 || D == 16 || D == 26;
}

int mobgs_raster_path(int total_channels, int class_filter, int n_tiles, const MobgsTuning* tuning) {
    const int D = total_channels;
    int bwd = 0;
    if (!class_filter && tuning_bwd_block_walk(tuning) && D >= 7 && D <= 10) {
        bwd = 3;
    } else if (tuning_bwd_mfma(tuning, n_tiles)) {
        const bool has = class_filter ? (D == 1 || D == 10) : (D == 1 || D == 3 || D == 4 || D == 9 || D == 10);
        if (has) bwd = tuning_bwd_mfma(tuning, n_tiles);
    }
    const bool fwd_blocks = tuning_block_walk(tuning) && (class_filter ? D == 10 : (D >= 7 && D <= 12));
    const int heavy_len = tuning_heavy_len(tuning, n_tiles);
    return bwd | (fwd_blocks ? 4 : 0) | (heavy_len > 0 ? 8 : 0) | (heavy_len << 8);
}

int mobgs_pack_records(int C, int N, int channels, const float* means2d, const float* conics, const float* colors,
                       int colors_per_camera, const float* opacities, int opac_per_camera, const float* extra,
                       const int32_t* radii, float* records, void* stream) {
    const int D = channels + (extra ? 1 : 0);
    if (C <= 0 || N < 0 || channels < 0 || D < 1) {
        set_error("mobgs_pack_records: bad sizes C=%d N=%d channels=%d", C, N, channels);
        return MOBGS_E_INVALID;
    }
    if (N == 0) return MOBGS_OK;
    hipLaunchKernelGGL(pack_records_kernel, dim3((N + 255) / 256, C), dim3(256), 0, (hipStream_t)stream, N, channels,
                       record_stride(D), means2d, conics, colors, colors_per_camera, opacities, opac_per_camera, extra,
                       radii, records);
    return check_launch("pack_records_kernel");
}

static int raster_fwd_impl(int C, int N, int channels, int width, int height, const float* means2d,
                           const float* conics, const float* colors, int colors_per_camera, const float* opacities,
                           int opac_per_camera, const float* extra, const float* backgrounds, const int32_t* radii,
                           const int32_t* tile_offsets, const int32_t* tile_order, const int32_t* flatten_ids,
                           float* records, float* render, float* alphas, int32_t* last_ids, uint8_t* isect_reach,
                           const MobgsTuning* tuning, void* stream, const DecodeEpi* dec) {
    hipStream_t st = (hipStream_t)stream;
    const int g_all_reach = tuning_all_reach(tuning);
    const int D = channels + (extra ? 1 : 0);
    if (C <= 0 || N < 0 || channels < 0 || D < 1 || width <= 0 || height <= 0) {
        set_error("mobgs_raster_fwd: bad sizes C=%d N=%d channels=%d W=%d H=%d", C, N, channels, width, height);
        return MOBGS_E_INVALID;
    }
    const int tile_w = (width + MOBGS_TILE - 1) / MOBGS_TILE, tile_h = (height + MOBGS_TILE - 1) / MOBGS_TILE;
    const int nt = C * tile_w * tile_h;
    const int stride = record_stride(D);
    if (N > 0 && colors != nullptr) {  // colors == NULL: `records` were packed by mobgs_project_and_bin_speculative
        hipLaunchKernelGGL(pack_records_kernel, dim3((N + 255) / 256, C), dim3(256), 0, st, N, channels, stride,
                           means2d, conics, colors, colors_per_camera, opacities, opac_per_camera, extra, radii,
                           records);
    }
    const int n_groups = (nt + TILES_PER_WG - 1) / TILES_PER_WG;
    const int grid = tile_order ? (int)((sched_slots((size_t)nt) + TILES_PER_WG - 1) / TILES_PER_WG) : ((n_groups + 7) / 8) * 8;
    const int block_walk = tuning_block_walk(tuning);
    if (dec) {  // decoder epilogue: the block-walk kernel of the 9 + 1 channel pass
        if (D != 10 || !extra || !block_walk) {
            set_error("mobgs_raster_fwd_decode: needs 9 feature channels + the depth channel and the block-walk kernel");
            return MOBGS_E_UNSUPPORTED;
        }
        hipLaunchKernelGGL((raster_fwd_blocks_kernel<10, false, true>), dim3(grid), dim3(64 * TILES_PER_WG), 0, st, nt,
                           n_groups, tile_w, tile_h, width, height, records, backgrounds, tile_offsets, flatten_ids, render,
                           alphas, last_ids, tile_order, ClassSel{0, 1, 0, g_all_reach}, isect_reach, *dec);
        return check_launch("raster_fwd_kernel(decode)");
    }
    const int rc = dispatch_channels(D, [&](auto cd) {
        constexpr int CD = decltype(cd)::value;
        // measured (profiles/r03): the block walk wins where a pixel's blend is wide -- 10 channels 256 -> 218 us, 12
        // channels 280 -> 255 us -- and loses where the per-step bookkeeping dominates (1 channel 86 -> 90 us) or the
        // accumulators leave two waves per SIMD (16 channels 313 -> 335 us)
        if (block_walk && CD >= 7 && CD <= 12)
            hipLaunchKernelGGL((raster_fwd_blocks_kernel<CD, false>), dim3(grid), dim3(64 * TILES_PER_WG), 0, st, nt,
                               n_groups, tile_w, tile_h, width, height, records, backgrounds, tile_offsets, flatten_ids,
                               render, alphas, last_ids, tile_order, ClassSel{0, 1, 0, g_all_reach}, isect_reach, DecodeEpi{});
        else
        hipLaunchKernelGGL((raster_fwd_kernel<CD, false>), dim3(grid), dim3(64 * TILES_PER_WG), 0, st, nt, n_groups,
                           tile_w, tile_h, width, height, records, backgrounds, tile_offsets, flatten_ids, render,
                           alphas, last_ids, tile_order, ClassSel{0, 1, 0, g_all_reach}, isect_reach);
    });
    if (rc != MOBGS_OK) {
        set_error("mobgs_raster_fwd: %d total channels not compiled in (pad to a supported count)", D);
        return rc;
    }
    return check_launch("raster_fwd_kernel");
}

int mobgs_raster_fwd(int C, int N, int channels, int width, int height, const float* means2d,
                     const float* conics, const float* colors, int colors_per_camera, const float* opacities,
                     int opac_per_camera, const float* extra, const float* backgrounds, const int32_t* radii,
                     const int32_t* tile_offsets, const int32_t* tile_order, const int32_t* flatten_ids,
                     float* records, float* render, float* alphas, int32_t* last_ids, uint8_t* isect_reach,
                     const MobgsTuning* tuning, void* stream) {
    return raster_fwd_impl(C, N, channels, width, height, means2d, conics, colors, colors_per_camera, opacities,
                           opac_per_camera, extra, backgrounds, radii, tile_offsets, tile_order, flatten_ids, records, render,
                           alphas, last_ids, isect_reach, tuning, stream, nullptr);
}

int mobgs_raster_fwd_decode(int C, int N, int channels, int width, int height, const float* means2d,
                            const float* conics, const float* colors, int colors_per_camera, const float* opacities,
                            int opac_per_camera, const float* extra, const float* backgrounds, const int32_t* radii,
                            const int32_t* tile_offsets, const int32_t* tile_order, const int32_t* flatten_ids,
                            float* records, float* render, float* alphas, int32_t* last_ids, uint8_t* isect_reach,
                            const float* ray_intr, int intr_stride, const float* ray_c2w, int c2w_stride, const float* w1,
                            const float* w2, float* rgb, float* depth, const MobgsTuning* tuning, void* stream) {
    if (!ray_intr || !ray_c2w || !w1 || !w2 || !rgb || !depth) {
        set_error("mobgs_raster_fwd_decode: ray_intr, ray_c2w, w1, w2, rgb and depth are required");
        return MOBGS_E_INVALID;
    }
    const DecodeEpi dec{ray_intr, ray_c2w, w1, w2, rgb, depth, intr_stride, c2w_stride};
    return raster_fwd_impl(C, N, channels, width, height, means2d, conics, colors, colors_per_camera, opacities,
                           opac_per_camera, extra, backgrounds, radii, tile_offsets, tile_order, flatten_ids, records, render,
                           alphas, last_ids, isect_reach, tuning, stream, &dec);
}

static int raster_bwd_impl(int C, int N, int channels, int has_extra, int width, int height, const float* records,
                     const float* backgrounds, const int32_t* radii, const float* means2d,
                     const int32_t* cum_tiles, const int32_t* keep_scan, const int32_t* tile_offsets,
                     const int32_t* tile_order, const int32_t* flatten_ids, const float* render_alphas,
                     const int32_t* last_ids, const float* v_render, const float* v_alphas, float* grad_slots,
                     const uint8_t* isect_reach, int32_t* any_record, const MobgsTuning* tuning, void* stream,
                     const DecodeBwd* db) {
    hipStream_t st = (hipStream_t)stream;
    const int g_all_reach = tuning_all_reach(tuning);
    (void)means2d;
    const int D = channels + (has_extra ? 1 : 0);
    if (C <= 0 || N < 0 || D < 1) {
        set_error("mobgs_raster_bwd: bad sizes C=%d N=%d channels=%d", C, N, channels);
        return MOBGS_E_INVALID;
    }
    const int tile_w = (width + MOBGS_TILE - 1) / MOBGS_TILE, tile_h = (height + MOBGS_TILE - 1) / MOBGS_TILE;
    const int nt = C * tile_w * tile_h;
    const int stride = record_stride(D);
    const int n_groups = (nt + TILES_PER_WG - 1) / TILES_PER_WG;
    const int grid = tile_order ? (int)((sched_slots((size_t)nt) + TILES_PER_WG - 1) / TILES_PER_WG) : ((n_groups + 7) / 8) * 8;
    const int bwd_blocks = tuning_bwd_block_walk(tuning);
    ClassSel cls{0, 1, 0, g_all_reach};
    cls.static_rows = tuning_static_rows(tuning);
    cls.set_n = N > 0 ? N : 1;
    cls.cover = tuning_cover_slots(tuning);
    const bool quadrant_selected = (mobgs_raster_path(D, 0, nt, tuning) & 3) == 0;   // (what the dispatch below arrives at)
    if (cls.cover && (!quadrant_selected || tuning_gate_zero_cotangent(tuning))) {
        set_error("mobgs_raster_bwd: cover_slots needs the quadrant kernel and no zero-cotangent gate");
        return MOBGS_E_UNSUPPORTED;
    }
    if (db) {   // decoder prologue: the quadrant kernel of the 9 + 1 channel pass (mobgs_raster_path tells beforehand)
        if (D != 10 || !has_extra || !quadrant_selected) {
            set_error("mobgs_raster_bwd_decode: needs 9 feature channels + the depth channel and the quadrant kernel "
                      "(mobgs_raster_path(10, 0, n_tiles, tuning) & 3 == 0)");
            return MOBGS_E_UNSUPPORTED;
        }
        hipLaunchKernelGGL((raster_bwd_kernel<10, false, true>), dim3(grid), dim3(64 * TILES_PER_WG), 0, st, nt, n_groups,
                           tile_w, tile_h, width, height, records, backgrounds, radii, cum_tiles, keep_scan,
                           tile_offsets, flatten_ids, render_alphas, last_ids, v_render, v_alphas, grad_slots,
                           tile_order, cls, isect_reach, any_record, *db);
        return check_launch("raster_bwd_kernel(decode)");
    }
    if (!bwd_blocks)
        cls.gate = arm_cotangent_gate(tuning, v_render, (size_t)C * height * width * D, v_alphas,
                                      (size_t)C * height * width, any_record, st);
    if (tuning_bwd_mfma(tuning, nt) && !bwd_blocks &&
        raster_bwd_mfma_launch(tuning_bwd_mfma(tuning, nt), D, false, grid, st, nt, n_groups, tile_w, tile_h, width, height, records, backgrounds,
                               radii, cum_tiles, keep_scan, tile_offsets, flatten_ids, render_alphas, last_ids, v_render,
                               v_alphas, grad_slots, tile_order, cls, isect_reach, any_record))
        return check_launch("raster_bwd_mfma_kernel");
    const int rc = dispatch_channels(D, [&](auto cd) {
        constexpr int CD = decltype(cd)::value;
        if constexpr (CD >= 7 && CD <= 10) {
            if (bwd_blocks) {
                hipLaunchKernelGGL((raster_bwd_blocks_kernel<CD>), dim3(grid), dim3(64 * TILES_PER_WG), 0, st, nt,
                                   n_groups, tile_w, tile_h, width, height, records, backgrounds, radii, cum_tiles,
                                   keep_scan, tile_offsets, flatten_ids, render_alphas, last_ids, v_render, v_alphas,
                                   grad_slots, tile_order, g_all_reach, isect_reach, any_record);
                return;
            }
        }
        hipLaunchKernelGGL((raster_bwd_kernel<CD, false>), dim3(grid), dim3(64 * TILES_PER_WG), 0, st, nt, n_groups,
                           tile_w, tile_h, width, height, records, backgrounds, radii, cum_tiles, keep_scan,
                           tile_offsets, flatten_ids, render_alphas, last_ids, v_render, v_alphas, grad_slots,
                           tile_order, cls, isect_reach, any_record, DecodeBwd{});
    });
    if (rc != MOBGS_OK) {
        set_error("mobgs_raster_bwd: %d total channels not compiled in", D);
        return rc;
    }
    return check_launch("raster_bwd_kernel");
}

int mobgs_raster_bwd(int C, int N, int channels, int has_extra, int width, int height, const float* records,
                     const float* backgrounds, const int32_t* radii, const float* means2d,
                     const int32_t* cum_tiles, const int32_t* keep_scan, const int32_t* tile_offsets,
                     const int32_t* tile_order, const int32_t* flatten_ids, const float* render_alphas,
                     const int32_t* last_ids, const float* v_render, const float* v_alphas, float* grad_slots,
                     const uint8_t* isect_reach, int32_t* any_record, const MobgsTuning* tuning, void* stream) {
    return raster_bwd_impl(C, N, channels, has_extra, width, height, records, backgrounds, radii, means2d, cum_tiles,
                           keep_scan, tile_offsets, tile_order, flatten_ids, render_alphas, last_ids, v_render, v_alphas,
                           grad_slots, isect_reach, any_record, tuning, stream, nullptr);
}

int mobgs_raster_bwd_decode(int C, int N, int width, int height, const float* records, const float* backgrounds,
                            const int32_t* radii, const int32_t* cum_tiles, const int32_t* keep_scan,
                            const int32_t* tile_offsets, const int32_t* tile_order, const int32_t* flatten_ids,
                            const float* render, const float* render_alphas, const int32_t* last_ids, const float* v_rgb,
                            const float* v_depth, const float* v_alphas, const float* ray_intr, int intr_stride,
                            const float* ray_c2w, int c2w_stride, const float* w1, const float* w2, float* grad_slots,
                            const uint8_t* isect_reach, int32_t* any_record, float* w_partial, const MobgsTuning* tuning,
                            void* stream) {
    if (!render || !v_rgb || !ray_intr || !ray_c2w || !w1 || !w2 || !w_partial) {
        set_error("mobgs_raster_bwd_decode: render, v_rgb, ray_intr, ray_c2w, w1, w2 and w_partial are required");
        return MOBGS_E_INVALID;
    }
    const int tiles = ((width + MOBGS_TILE - 1) / MOBGS_TILE) * ((height + MOBGS_TILE - 1) / MOBGS_TILE);
    const DecodeBwd db{render, v_rgb, v_depth, ray_intr, ray_c2w, w1, w2, w_partial, decoder_wgrad_ticket(w_partial, C, tiles),
                       intr_stride, c2w_stride};
    return raster_bwd_impl(C, N, 9, 1, width, height, records, backgrounds, radii, nullptr, cum_tiles, keep_scan,
                           tile_offsets, tile_order, flatten_ids, render_alphas, last_ids, nullptr, v_alphas,
                           grad_slots, isect_reach, any_record, tuning, stream, &db);
}

int mobgs_raster_bwd_decode_finish(int C, int width, int height, float* w_partial, int c2w_stride, float* g_w1,
                                   float* g_w2, float* g_c2w, int g_c2w_floats, int accumulate_wgrad, void* stream) {
    if (C <= 0 || width <= 0 || height <= 0 || !w_partial || !g_w1 || !g_w2 ||
        (g_c2w && g_c2w_floats != 12 && g_c2w_floats != 16)) {
        set_error("mobgs_raster_bwd_decode_finish: w_partial, g_w1 and g_w2 are required (g_c2w: 12 or 16 floats per image)");
        return MOBGS_E_INVALID;
    }
    if (C > 1 && g_c2w && c2w_stride == 0) {
        set_error("mobgs_raster_bwd_decode_finish: a pose shared by the images of a batch cannot receive a gradient");
        return MOBGS_E_INVALID;
    }
    const int tiles = ((width + MOBGS_TILE - 1) / MOBGS_TILE) * ((height + MOBGS_TILE - 1) / MOBGS_TILE);
    launch_decoder_wgrad_reduce(C, tiles, w_partial, g_w1, g_w2, g_c2w, g_c2w_floats, accumulate_wgrad, (hipStream_t)stream);
    return check_launch("decoder_wgrad_reduce_rows_kernel");
}

int mobgs_raster_bwd_reduce_decode(int C, int N, const float* records, const int32_t* cum_tiles, const int32_t* keep_scan,
                                   const float* grad_slots, const int32_t* any_record, float* v_means2d, float* v_conics,
                                   float* v_opacities, float* v_colors, float* v_extra, const int32_t* tiles_per_gauss,
                                   int width, int height, const float* w_partial, int c2w_stride, float* g_w1, float* g_w2,
                                   float* g_c2w, int g_c2w_floats, int accumulate_wgrad, void* stream) {
    if (C <= 0 || N < 0 || width <= 0 || height <= 0 || ((long long)C * N > 0 && !records) || !w_partial || !g_w1 || !g_w2 ||
        (g_c2w && g_c2w_floats != 12 && g_c2w_floats != 16) || (C > 1 && g_c2w && c2w_stride == 0)) {
        set_error("mobgs_raster_bwd_reduce_decode: bad arguments (w_partial, g_w1, g_w2 required; g_c2w: 12 or 16 floats per "
                  "image, one pose per image)");
        return MOBGS_E_INVALID;
    }
    WgradFinish wf;
    wf.w_partial = w_partial;
    wf.g_w1 = g_w1;
    wf.g_w2 = g_w2;
    wf.g_c2w = g_c2w;
    wf.rows_per_image = ((width + MOBGS_TILE - 1) / MOBGS_TILE) * ((height + MOBGS_TILE - 1) / MOBGS_TILE);
    wf.n_images = C;
    wf.accumulate = accumulate_wgrad;
    wf.c2w_floats = g_c2w_floats;
    const int n = C * N;
    const int extra = wgrad_finish_outputs(wf) * C;
    hipLaunchKernelGGL((slot_reduce16_kernel<8, true>), dim3(extra + (int)(((size_t)n * 8 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, n, 9, 1, cum_tiles, keep_scan, grad_slots, v_means2d, v_conics, v_opacities,
                       v_colors, v_extra, any_record, records, tiles_per_gauss, wf);
    return check_launch("slot_reduce16_kernel(+wgrad)");
}

size_t mobgs_raster_bwd_decode_scratch_floats(int C, int width, int height) {
    if (C <= 0 || width <= 0 || height <= 0) return 0;
    return decoder_wgrad_scratch_floats(C, ((width + MOBGS_TILE - 1) / MOBGS_TILE) * ((height + MOBGS_TILE - 1) / MOBGS_TILE));
}

// class-restricted passes over the lists of the whole set: 10 total channels (the render() configuration) or 1 (the
// dynamic-only coverage of get_flow(): only the alpha output matters there)
int mobgs_raster_class_fwd(int C, int N, int Ns, int class_sel, int channels_total, int width, int height,
                           const float* records, const float* backgrounds, const int32_t* tile_offsets,
                           const int32_t* tile_order, const int32_t* flatten_ids, float* render, float* alphas,
                           int32_t* last_ids, uint8_t* isect_reach, const MobgsTuning* tuning, void* stream) {
    const int g_all_reach = tuning_all_reach(tuning);
    if (C <= 0 || N <= 0 || Ns < 0 || Ns > N || (class_sel != 1 && class_sel != 2) ||
        (channels_total != 10 && channels_total != 1)) {
        set_error("mobgs_raster_class_fwd: unsupported arguments (C=%d N=%d Ns=%d class=%d D=%d)", C, N, Ns, class_sel,
                  channels_total);
        return MOBGS_E_UNSUPPORTED;
    }
    const int tile_w = (width + MOBGS_TILE - 1) / MOBGS_TILE, tile_h = (height + MOBGS_TILE - 1) / MOBGS_TILE;
    const int nt = C * tile_w * tile_h;
    const int n_groups = (nt + TILES_PER_WG - 1) / TILES_PER_WG;
    const int grid = tile_order ? (int)((sched_slots((size_t)nt) + TILES_PER_WG - 1) / TILES_PER_WG) : ((n_groups + 7) / 8) * 8;
    const ClassSel cls{class_sel, N, Ns, g_all_reach};
    const dim3 g3(grid), b3(64 * TILES_PER_WG);
    hipStream_t st = (hipStream_t)stream;
    if (tuning_block_walk(tuning) && channels_total == 10)  // (the 1-channel coverage pass: 86 us vs 90 us, see above)
        hipLaunchKernelGGL((raster_fwd_blocks_kernel<10, true>), g3, b3, 0, st, nt, n_groups, tile_w, tile_h, width,
                           height, records, backgrounds, tile_offsets, flatten_ids, render, alphas, last_ids,
                           tile_order, cls, isect_reach, DecodeEpi{});
    else if (channels_total == 10)
        hipLaunchKernelGGL((raster_fwd_kernel<10, true>), g3, b3, 0, st, nt, n_groups, tile_w, tile_h, width, height,
                           records, backgrounds, tile_offsets, flatten_ids, render, alphas, last_ids, tile_order, cls,
                           isect_reach);
    else
        hipLaunchKernelGGL((raster_fwd_kernel<1, true>), g3, b3, 0, st, nt, n_groups, tile_w, tile_h, width, height,
                           records, backgrounds, tile_offsets, flatten_ids, render, alphas, last_ids, tile_order, cls,
                           isect_reach);
    return check_launch("raster_fwd_kernel(class)");
}

int mobgs_raster_class_bwd(int C, int N, int Ns, int class_sel, int channels_total, int width, int height,
                           const float* records, const float* backgrounds, const int32_t* radii,
                           const int32_t* cum_tiles, const int32_t* keep_scan, const int32_t* tile_offsets,
                           const int32_t* tile_order, const int32_t* flatten_ids, const float* render_alphas,
                           const int32_t* last_ids, const float* v_render, const float* v_alphas, float* grad_slots,
                           const uint8_t* isect_reach, int32_t* any_record, const MobgsTuning* tuning, void* stream) {
    const int g_all_reach = tuning_all_reach(tuning);
    if (C <= 0 || N <= 0 || Ns < 0 || Ns > N || (class_sel != 1 && class_sel != 2) ||
        (channels_total != 10 && channels_total != 1)) {
        set_error("mobgs_raster_class_bwd: unsupported arguments");
        return MOBGS_E_UNSUPPORTED;
    }
    const int tile_w = (width + MOBGS_TILE - 1) / MOBGS_TILE, tile_h = (height + MOBGS_TILE - 1) / MOBGS_TILE;
    const int nt = C * tile_w * tile_h;
    const int n_groups = (nt + TILES_PER_WG - 1) / TILES_PER_WG;
    const int grid = tile_order ? (int)((sched_slots((size_t)nt) + TILES_PER_WG - 1) / TILES_PER_WG) : ((n_groups + 7) / 8) * 8;
    ClassSel cls{class_sel, N, Ns, g_all_reach};
    cls.static_rows = tuning_static_rows(tuning);
    cls.set_n = N;
    cls.gate = arm_cotangent_gate(tuning, v_render, (size_t)C * height * width * channels_total, v_alphas,
                                  (size_t)C * height * width, any_record, (hipStream_t)stream);
    if (tuning_bwd_mfma(tuning, nt) &&
        raster_bwd_mfma_launch(tuning_bwd_mfma(tuning, nt), channels_total, true, grid, (hipStream_t)stream, nt, n_groups, tile_w, tile_h, width,
                               height, records, backgrounds, radii, cum_tiles, keep_scan, tile_offsets, flatten_ids,
                               render_alphas, last_ids, v_render, v_alphas, grad_slots, tile_order,
                               cls, isect_reach, any_record))
        return check_launch("raster_bwd_mfma_kernel(class)");
    if (channels_total == 10)
        hipLaunchKernelGGL((raster_bwd_kernel<10, true>), dim3(grid), dim3(64 * TILES_PER_WG), 0, (hipStream_t)stream,
                           nt, n_groups, tile_w, tile_h, width, height, records, backgrounds, radii, cum_tiles,
                           keep_scan, tile_offsets, flatten_ids, render_alphas, last_ids, v_render, v_alphas,
                           grad_slots, tile_order, cls, isect_reach, any_record, DecodeBwd{});
    else
        hipLaunchKernelGGL((raster_bwd_kernel<1, true>), dim3(grid), dim3(64 * TILES_PER_WG), 0, (hipStream_t)stream,
                           nt, n_groups, tile_w, tile_h, width, height, records, backgrounds, radii, cum_tiles,
                           keep_scan, tile_offsets, flatten_ids, render_alphas, last_ids, v_render, v_alphas,
                           grad_slots, tile_order, cls, isect_reach, any_record, DecodeBwd{});
    return check_launch("raster_bwd_kernel(class)");
}

int mobgs_raster_bwd_reduce(int C, int N, int channels, int has_extra, const float* records,
                            const int32_t* cum_tiles, const int32_t* keep_scan, const float* grad_slots,
                            const int32_t* any_record, float* v_means2d, float* v_conics, float* v_opacities,
                            float* v_colors, float* v_extra, const int32_t* tiles_per_gauss, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int D = channels + (has_extra ? 1 : 0);
    if (C <= 0 || N < 0 || D < 1 || ((long long)C * N > 0 && !records)) {
        set_error("mobgs_raster_bwd_reduce: bad sizes C=%d N=%d channels=%d", C, N, channels);
        return MOBGS_E_INVALID;
    }
    const int stride = record_stride(D);
    const int n = C * N;
    if (n > 0) {
        if (stride == 8) {
            hipLaunchKernelGGL(slot_reduce_wide_kernel<2>, dim3((int)(((size_t)n * 2 + 255) / 256)), dim3(256), 0, st, n,
                               channels, has_extra, 2, cum_tiles, keep_scan, grad_slots, v_means2d, v_conics,
                               v_opacities, v_colors, v_extra, any_record, records, tiles_per_gauss);
        } else if (stride < 8) {
            hipLaunchKernelGGL(slot_reduce_kernel<8>, dim3((n * 8 + 255) / 256), dim3(256), 0, st, n, channels,
                               has_extra, stride, cum_tiles, keep_scan, grad_slots, v_means2d, v_conics, v_opacities, v_colors,
                               v_extra, any_record, records, tiles_per_gauss);
        } else if (stride == 12) {
            hipLaunchKernelGGL(slot_reduce_wide_kernel<4>, dim3((int)(((size_t)n * 4 + 255) / 256)), dim3(256), 0, st, n,
                               channels, has_extra, 3, cum_tiles, keep_scan, grad_slots, v_means2d, v_conics,
                               v_opacities, v_colors, v_extra, any_record, records, tiles_per_gauss);
        } else if (stride == 16) {  // (slot_reduce_wide_kernel<4> measures the same here: 29.0 vs 28.7 us)
            hipLaunchKernelGGL(slot_reduce16_kernel<8>, dim3((int)(((size_t)n * 8 + 255) / 256)), dim3(256), 0, st, n,
                               channels, has_extra, cum_tiles, keep_scan, grad_slots, v_means2d, v_conics, v_opacities,
                               v_colors, v_extra, any_record, records, tiles_per_gauss);
        } else if (stride <= 16) {
            hipLaunchKernelGGL(slot_reduce_kernel<16>, dim3((int)(((size_t)n * 16 + 255) / 256)), dim3(256), 0, st, n,
                               channels, has_extra, stride, cum_tiles, keep_scan, grad_slots, v_means2d, v_conics, v_opacities,
                               v_colors, v_extra, any_record, records, tiles_per_gauss);
        } else if (stride <= 32 && (stride & 3) == 0) {
            hipLaunchKernelGGL(slot_reduce_wide_kernel<8>, dim3((int)(((size_t)n * 8 + 255) / 256)), dim3(256), 0, st, n,
                               channels, has_extra, stride / 4, cum_tiles, keep_scan, grad_slots, v_means2d, v_conics,
                               v_opacities, v_colors, v_extra, any_record, records, tiles_per_gauss);
        } else {
            hipLaunchKernelGGL(slot_reduce_kernel<32>, dim3((int)(((size_t)n * 32 + 255) / 256)), dim3(256), 0, st, n,
                               channels, has_extra, stride, cum_tiles, keep_scan, grad_slots, v_means2d, v_conics, v_opacities,
                               v_colors, v_extra, any_record, records, tiles_per_gauss);
        }
    }
    return check_launch("slot_reduce_kernel");
}

}  // extern "C"
