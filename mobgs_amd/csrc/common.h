// Shared helpers for the libmobgs_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mobgs_hip.h"

#define MOBGS_WAVE 64

namespace mobgs {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MOBGS_E_LAUNCH;
    }
    return MOBGS_OK;
}

__host__ __device__ inline int record_stride(int channels) { return (6 + channels + 3) & ~3; }

// XCD-aware remap of a linear workgroup id.  The dispatcher places workgroup b on XCD b % 8
// (MI355X_MICROARCH.md, "Workgroup dispatch"); giving each XCD one contiguous chunk of the logical
// index space keeps neighbouring tiles (which share splats) behind the same 4 MiB L2.  Only a
// performance hint: any placement gives the same result.
__device__ inline int xcd_chunked(int b, int n) {
    const int per = (n + 7) >> 3;
    const int logical = (b & 7) * per + (b >> 3);
    return logical;  // may be >= n for the ragged tail; callers bounds-check
}

// Tile rectangle [x0,x1) x [y0,y1) touched by a splat (gsplat isect_tiles.cu [upstream]).
struct TileRect {
    int x0, y0, x1, y1;
};
__host__ __device__ inline TileRect tile_rect(float mx, float my, int radius, int tile_w, int tile_h) {
    const float inv = 1.0f / (float)MOBGS_TILE;
    const float tr = (float)radius * inv;
    const float tx = mx * inv, ty = my * inv;
    TileRect r;
    float fx0 = floorf(tx - tr), fx1 = ceilf(tx + tr), fy0 = floorf(ty - tr), fy1 = ceilf(ty + tr);
    // clamp in float first so huge coordinates cannot overflow the int conversion
    fx0 = fminf(fmaxf(fx0, 0.f), (float)tile_w);
    fx1 = fminf(fmaxf(fx1, 0.f), (float)tile_w);
    fy0 = fminf(fmaxf(fy0, 0.f), (float)tile_h);
    fy1 = fminf(fmaxf(fy1, 0.f), (float)tile_h);
    r.x0 = (int)fx0;
    r.x1 = (int)fx1;
    r.y0 = (int)fy0;
    r.y1 = (int)fy1;
    return r;
}


// ---- tile schedule (tile_scan_kernel) ------------------------------------------------------------------------------
// tile_order has sched_slots(n_tiles) entries, 4 per workgroup of the compositing kernels: a tile id, a tile id |
// SCHED_HEAVY (in all 4 slots of one workgroup: the 4 waves share that tile, one 8x8 quadrant each) or -1 (unused).
// At most an eighth of the tiles (the longest) are scheduled heavy: when more lists than that are long, long is
// the norm and no single list is the critical path.  Small images are the exception: with no more tiles than the
// chip has SIMDs (1024) one wave per tile leaves SIMDs idle, exposes every latency and makes every list a critical
// path, so ALL tiles may be heavy there -- 576 at 512x288, the reference's own training resolution (raster_fwd 128 ->
// 66 us, raster_bwd 215 -> 113 us on a 30 k-splat scene; a 920-tile grid still gains 1.34x per step, a 1400-tile
// grid already loses 2 %: four waves per tile pay the per-entry overhead four times).
constexpr int SCHED_HEAVY = 1 << 30;
#ifndef MOBGS_SCHED_SMALL_GRID
#define MOBGS_SCHED_SMALL_GRID 1024  // (experiments: a huge value makes every grid "small" = every tile may be heavy)
#endif
constexpr size_t SCHED_SMALL_GRID = MOBGS_SCHED_SMALL_GRID;
__host__ __device__ inline size_t sched_max_heavy(size_t n_tiles) {
    return n_tiles <= SCHED_SMALL_GRID ? n_tiles : n_tiles / 8;
}
__host__ __device__ inline size_t sched_slots(size_t n_tiles) { return n_tiles + 3 * sched_max_heavy(n_tiles) + 4; }

// ---- compact index of a bounding-box intersection (= its gradient slot) -----------------------------------------
// keep_scan is stored in chunks of KEEP_CHUNK intersections, each preceded by one word: [base_c | local_0 ..
// local_2047] with local_i = number of kept intersections before i inside the chunk and base_c = number kept in
// all earlier chunks, so the chunks can be scanned independently and the bases filled in afterwards.
constexpr int KEEP_CHUNK_LOG2 = 11;
constexpr int KEEP_CHUNK = 1 << KEEP_CHUNK_LOG2;
__host__ __device__ inline size_t keep_scan_len(size_t capacity) {
    return ((capacity >> KEEP_CHUNK_LOG2) + 1) * (size_t)(KEEP_CHUNK + 1);
}
__device__ inline int keep_index(const int32_t* __restrict__ keep_scan, int j) {
    const int32_t* p = keep_scan + (size_t)(j >> KEEP_CHUNK_LOG2) * (KEEP_CHUNK + 1);
    return p[0] + p[1 + (j & (KEEP_CHUNK - 1))];
}

// ---- reach tests (bit-exact culling of work the compositor would skip anyway) --------------------------------
// Smallest sigma = 0.5 (a dx^2 + c dy^2) + b dx dy a splat can take over a pixel-centre rectangle
// (convex quadratic: 0 if the centre is inside, else attained on one of the four edges).
// sx = cb / cc and sy = cb / ca are constants of the splat: the fused binning path (isect.hip, bin_kernel<.., true>)
// reads them (and the halved conic diagonal) from the per-splat bin record the projection kernel wrote, the classic path
// computes them per intersection -- the same arithmetic either way, so both paths take the same decisions.
// ha = a / 2, hc = c / 2.  Every operation is written out (v_med3 clamps, explicit FMAs): this runs once per bounding-box
// intersection, 3.3 M times per frame on the benchmark, and bin_kernel is bound by its instruction count.
__device__ __forceinline__ float half_sigma_at(float ha, float cb, float hc, float dx, float dy) {
    // a dx^2 / 2 + b dx dy + c dy^2 / 2 = dx (ha dx + b dy) + (hc dy) dy
    return __fmaf_rn(hc * dy, dy, dx * __fmaf_rn(cb, dy, ha * dx));
}
__device__ inline float min_sigma_over_tile_pre(float mx, float my, float ha, float cb, float hc, float sx, float sy,
                                                float x0, float x1, float y0, float y1) {
    const bool inside = mx >= x0 && mx <= x1 && my >= y0 && my <= y1;
    float s[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        {  // vertical edge px = x0 / x1: optimum dy = -cb dx / cc, clamped to the edge
            const float dx = mx - (k ? x1 : x0);
            const float dy = my - __builtin_amdgcn_fmed3f(__fmaf_rn(sx, dx, my), y0, y1);
            s[2 * k] = half_sigma_at(ha, cb, hc, dx, dy);
        }
        {  // horizontal edge py = y0 / y1
            const float dy = my - (k ? y1 : y0);
            const float dx = mx - __builtin_amdgcn_fmed3f(__fmaf_rn(sy, dy, mx), x0, x1);
            s[2 * k + 1] = half_sigma_at(ha, cb, hc, dx, dy);
        }
    }
    const float best = fminf(fminf(s[0], s[1]), fminf(s[2], s[3]));
    return inside ? 0.f : best;
}
__device__ inline float min_sigma_over_tile(float mx, float my, float ca, float cb, float cc, float x0, float x1,
                                            float y0, float y1) {
    // two divisions for the four edges (as quadrant_reach_mask below)
    return min_sigma_over_tile_pre(mx, my, 0.5f * ca, cb, 0.5f * cc, cb / cc, cb / ca, x0, x1, y0, y1);
}
// alpha = min(0.999, op * exp(-sigma)) >= 1/255  <=>  sigma <= ln(255 op).  The returned threshold carries a
// conservative margin (the compositor evaluates sigma at pixel centres in a different fp32 operation order);
// NEGATIVE when the splat can never reach 1/255.
__device__ inline float reach_threshold(float op) {
    if (!(op * 255.f >= 1.f)) return -1.f;
    const float tau = __logf(255.f * op);
    // (operation order pinned: the projection kernel -- built without FMA contraction -- evaluates this for the bin
    // records of the fused path and must agree bit for bit with the binning / compositing translation units)
    return __fmaf_rn(0.02f, tau, __fadd_rn(tau, 0.05f));
}

// ---- per-splat bin record (fused binning path) -----------------------------------------------------------------
// Everything bin_kernel needs to know about a visible splat, in ONE 48-byte row written by the projection kernel while
// the values are in registers (instead of five gathers from means2d / radii / conics / opacities / depths and two
// divisions + a logarithm per bounding-box intersection):
//   q0 = {mean2d.x, mean2d.y, reach threshold, depth bits}    threshold: -1 = the splat never reaches alpha >= 1/255,
//                                                             3e38 = list it in every tile of its box (culling off,
//                                                             or a degenerate conic)
//   q1 = {conic a / 2, b, c / 2, first tile of the box: x0 | y0 << 16}
//   q2 = {b / c, b / a, box width in tiles | camera << 16, unused}
constexpr int BIN_RECORD_FLOATS = 12;
constexpr float REACH_ALWAYS = 3.0e38f;
__device__ inline void write_bin_record(float* __restrict__ r, float mx, float my, float ca, float cb, float cc,
                                        float depth, float op, int cull, const TileRect& tr, int cam) {
    float thr = REACH_ALWAYS;
    if (cull) {
        if (!(op * 255.f >= 1.f))
            thr = -1.f;
        else if (ca > 0.f && cc > 0.f)
            thr = reach_threshold(op);
    }
    float4* o = reinterpret_cast<float4*>(r);
    o[0] = make_float4(mx, my, thr, depth);
    o[1] = make_float4(0.5f * ca, cb, 0.5f * cc, __uint_as_float((unsigned)tr.x0 | ((unsigned)tr.y0 << 16)));
    o[2] = make_float4(cb / cc, cb / ca, __uint_as_float((unsigned)(tr.x1 - tr.x0) | ((unsigned)cam << 16)), 0.f);
}
// bit k set <=> the splat can reach alpha >= 1/255 somewhere in rows [4k, 4k+3] of the 16x16 tile (tx, ty)
__device__ inline int band_mask(float mx, float my, float ca, float cb, float cc, float op, int tx, int ty, int width,
                                int height) {
    const float thr = reach_threshold(op);
    if (thr < 0.f) return 0;
    if (!(ca > 0.f && cc > 0.f)) return 0xF;
    const float x0 = (float)(tx * MOBGS_TILE) + 0.5f;
    const float x1 = fminf((float)(tx * MOBGS_TILE) + 15.5f, (float)width - 0.5f);
    int m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float y0 = (float)(ty * MOBGS_TILE + 4 * k) + 0.5f;
        if (y0 > (float)height) break;
        const float y1 = fminf(y0 + 3.f, (float)height - 0.5f);
        if (min_sigma_over_tile(mx, my, ca, cb, cc, x0, x1, y0, y1) <= thr) m |= 1 << k;
    }
    return m;
}

// The compositor's packed splat record {x, y, A, B | C, L, colour[0..1] | colour ... (extra) 0...}: `stride` floats (a
// multiple of 4), 16-byte aligned.  Conic and opacity are stored in EXPONENT FORM: with the conic (a, b, c) and
// d = mean2d - pixel,
//     opacity * exp(-sigma) = exp2(A dx^2 + C dy^2 + B dx dy + L),
//     A = -log2(e) a / 2,  B = -log2(e) b,  C = -log2(e) c / 2,  L = log2(opacity)
// -- the per-(pixel, splat) evaluation (eval_splat, raster_shared.h) is then three multiplies, three FMAs and the
// hardware exp2, without the scaling of sigma and the multiplication by the opacity (3 VALU fewer per pair in every
// compositing kernel).  sigma < 0 (upstream's skip test) reads "exponent > L".  The few consumers that need conic and
// opacity themselves (reach masks per staged entry, the per-splat finish of the gradient reduction) convert back.
constexpr float MOBGS_LOG2E = 1.4426950408889634f;
constexpr float MOBGS_LN2 = 0.6931471805599453f;
__device__ inline void record_exponent_form(float ca, float cb, float cc, float op, float& A, float& B, float& C,
                                            float& L) {
    A = (-0.5f * MOBGS_LOG2E) * ca;
    B = -MOBGS_LOG2E * cb;
    C = (-0.5f * MOBGS_LOG2E) * cc;
    L = op > 0.f ? __log2f(op) : -__builtin_inff();  // (opacity <= 0 or NaN: the splat never reaches 1/255)
}
__device__ inline void record_conic_form(float A, float B, float C, float L, float& ca, float& cb, float& cc,
                                         float& op) {
    ca = (-2.f * MOBGS_LN2) * A;
    cb = -MOBGS_LN2 * B;
    cc = (-2.f * MOBGS_LN2) * C;
    op = __builtin_amdgcn_exp2f(L);
}
__device__ inline void write_splat_record(float* __restrict__ r, float x, float y, float ca, float cb, float cc,
                                          float op, const float* __restrict__ col, int channels, bool has_extra,
                                          float extra) {
    float A, B, C, L;
    record_exponent_form(ca, cb, cc, op, A, B, C, L);
    reinterpret_cast<float4*>(r)[0] = make_float4(x, y, A, B);
    const int D = channels + (has_extra ? 1 : 0);
    float buf[4] = {C, L, 0.f, 0.f};
    int fill = 2;
    int q = 1;
    for (int k = 0; k < D; ++k) {
        buf[fill++] = (k < channels) ? col[k] : extra;
        if (fill == 4) {
            reinterpret_cast<float4*>(r)[q++] = make_float4(buf[0], buf[1], buf[2], buf[3]);
            fill = 0;
            buf[0] = buf[1] = buf[2] = buf[3] = 0.f;
        }
    }
    if (fill > 0) reinterpret_cast<float4*>(r)[q++] = make_float4(buf[0], buf[1], buf[2], buf[3]);
}
// optional side job of project_fwd: pack the records with the depth as extra channel (records == NULL: off)
struct PackArgs {
    const float* opacities;
    const float* colors;
    float* records;
    int opac_per_camera, colors_per_camera, channels, stride;
};

// optional side job of project_fwd: the per-splat bin records of the fused binning path (records == NULL: off)
struct BinArgs {
    float* records;           // [C*N, BIN_RECORD_FLOATS]
    const float* opacities;   // [N] or [C,N]
    int opac_per_camera, cull;
};

// ---- launchers shared between translation units (the orchestrator in pipeline.hip fuses small steps) ---------
// project_fwd with the option to clear `zero_n` ints at `zero_ptr` on the way (the binning scratch counters) and to
// pack the compositor's records
int project_fwd_launch(int C, int N, const float* means, const float* quats, const float* scales, const float* viewmats,
                       const float* Ks, int width, int height, float eps2d, float near_plane, float far_plane,
                       float radius_clip, int32_t* radii, float* means2d, float* depths, float* conics,
                       int32_t* tiles_per_gauss, int32_t* zero_ptr, size_t zero_n, PackArgs pack, void* stream,
                       int geometry_per_camera = 0, BinArgs bin = BinArgs{nullptr, nullptr, 0, 0},
                       const MobgsPrepInputs* prep = nullptr);
// mobgs_isect_offsets; scratch_zeroed: the counters were cleared by the caller; stats_mirror: device-visible host
// address that receives a copy of stats[0..2] (or NULL) and then, in word 3, stats_seq (when non-zero)
int isect_offsets_launch(int C, int N, int tile_w, int tile_h, int width, int height, int cull, int capacity,
                         const int32_t* tiles_per_gauss, const float* means2d, const int32_t* radii, const float* conics,
                         const float* opacities, int opac_per_camera, int32_t* cum_tiles, int32_t* keep_scan,
                         int32_t* tile_offsets, int32_t* tile_order, int64_t capacity_listed, int64_t* stats,
                         void* scratch, bool scratch_zeroed, int64_t* stats_mirror, int64_t stats_seq,
                         const MobgsTuning* tuning, void* stream);

// per-call policy (include/mobgs_hip.h MobgsTuning): NULL or a negative field = the library default
// default: lists of >= 1024 entries; on grids too small to fill the chip every non-empty list qualifies
inline int tuning_heavy_len(const MobgsTuning* t, int n_tiles) {
    if (t && t->heavy_tile_len >= 0) return t->heavy_tile_len;
    return (size_t)n_tiles <= SCHED_SMALL_GRID ? 1 : 1024;
}
inline int tuning_list_hint(const MobgsTuning* t) { return (t && t->longest_list_hint >= 0) ? t->longest_list_hint : 0; }
inline int tuning_all_reach(const MobgsTuning* t) { return (t && t->quadrant_culling == 0) ? 1 : 0; }
inline int tuning_block_walk(const MobgsTuning* t) { return (t && t->block_walk == 0) ? 0 : 1; }
inline int tuning_bwd_block_walk(const MobgsTuning* t) { return (t && t->bwd_block_walk == 1) ? 1 : 0; }
// backward compositor with the gradient sums on the matrix pipe (raster_bwd_mfma.hip; MobgsTuning.bwd_mfma): 0 = off,
// 1 = one wave per tile + the four-wave team for the schedule's heavy tiles, 2 = the team for every tile.  Library
// default (field < 0): 1 on grids of <= SCHED_SMALL_GRID tiles -- every tile is heavy there and the team kernel measures
// 92 us against 120 us (512x288, 30 k splats) -- and 0 on larger grids, where the quadrant kernel is faster (517 us
// against 614 / 722 us at 1352x1014, 300 k splats: DESIGN.md section 4d).
#ifndef MOBGS_BWD_MFMA_DEFAULT
#define MOBGS_BWD_MFMA_DEFAULT (-1)
#endif
inline int tuning_bwd_mfma(const MobgsTuning* t, int n_tiles) {
    int v = (t && t->bwd_mfma >= 0) ? t->bwd_mfma : MOBGS_BWD_MFMA_DEFAULT;
    if (v < 0) v = (size_t)n_tiles <= SCHED_SMALL_GRID ? 1 : 0;
    return v > 2 ? 2 : v;
}
inline int tuning_coherent_order(const MobgsTuning* t) { return (t && t->coherent_order == 1) ? 1 : 0; }
inline int tuning_gate_zero_cotangent(const MobgsTuning* t) { return (t && t->gate_zero_cotangent == 1) ? 1 : 0; }
inline int tuning_static_rows(const MobgsTuning* t) { return (t && t->static_rows > 0) ? t->static_rows : 0; }
inline int tuning_cover_slots(const MobgsTuning* t) { return (t && t->cover_slots > 0) ? 1 : 0; }
inline int tuning_geometry_per_camera(const MobgsTuning* t) { return (t && t->geometry_per_camera == 1) ? 1 : 0; }
void isect_zeroed_region(void* scratch, size_t n_gauss, size_t n_tiles, size_t capacity, int32_t** ptr, size_t* count);
// fused single-pass lists (isect.hip): where the projection kernel leaves the bin records inside the binning scratch,
// and the launcher of scan -> bin (keys into strided segments) -> offsets / schedule -> per-tile sort
float* isect_bin_records(void* scratch, size_t n_gauss, size_t n_tiles, size_t capacity);
int isect_fused_launch(int C, int N, int tile_w, int tile_h, int width, int height, int capacity,
                       const int32_t* tiles_per_gauss, int32_t* cum_tiles, int32_t* keep_scan, int32_t* tile_offsets,
                       int32_t* tile_order, int64_t capacity_listed, int64_t* stats, void* scratch, int64_t* stats_mirror,
                       int64_t stats_seq, uint64_t* seg_keys, int seg_stride, int32_t* flatten_ids, uint64_t* isect_ids,
                       int64_t max_tile_len_hint, const int32_t* enum_order, const MobgsTuning* tuning, void* stream);

// bit q = 2 * qy + qx set <=> the splat may reach alpha >= 1/255 at a pixel centre of the 8x8 quadrant (qx, qy) of
// the 16x16 tile (tx, ty).  Same conservative test as min_sigma_over_tile / reach_threshold, on the four quadrant
// rectangles at once: the 4 + 4 lines carrying their edges are each solved once (optimum of the convex quadratic
// along the line, clamped to the two segments).  Quadrants are NOT clipped to the image (only more conservative).
__device__ inline unsigned quadrant_reach_mask(float mx, float my, float ca, float cb, float cc, float op, int tx,
                                               int ty) {
    const float thr = reach_threshold(op);
    if (thr < 0.f) return 0u;
    if (!(ca > 0.f && cc > 0.f)) return 0xFu;
    const float x0 = (float)(tx * MOBGS_TILE) + 0.5f, y0 = (float)(ty * MOBGS_TILE) + 0.5f;
    const float sx = cb / cc, sy = cb / ca;
    float best[4] = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float off = (float)((i & 1) * 7 + (i >> 1) * 8);
        {  // vertical line px = x0 + off: an edge of the quadrants of column i >> 1
            const float dx = mx - (x0 + off);
            const float ystar = my + sx * dx;
#pragma unroll
            for (int qy = 0; qy < 2; ++qy) {
                const float ya = y0 + (float)(8 * qy);
                const float dy = my - fminf(fmaxf(ystar, ya), ya + 7.f);
                const float sg = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                best[2 * qy + (i >> 1)] = fminf(best[2 * qy + (i >> 1)], sg);
            }
        }
        {  // horizontal line py = y0 + off: an edge of the quadrants of row i >> 1
            const float dy = my - (y0 + off);
            const float xstar = mx + sy * dy;
#pragma unroll
            for (int qx = 0; qx < 2; ++qx) {
                const float xa = x0 + (float)(8 * qx);
                const float dx = mx - fminf(fmaxf(xstar, xa), xa + 7.f);
                const float sg = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                best[2 * (i >> 1) + qx] = fminf(best[2 * (i >> 1) + qx], sg);
            }
        }
    }
    unsigned m = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float xa = x0 + (float)(8 * (q & 1)), ya = y0 + (float)(8 * (q >> 1));
        const bool inside = mx >= xa && mx <= xa + 7.f && my >= ya && my <= ya + 7.f;
        if (inside || best[q] <= thr) m |= 1u << q;
    }
    return m;
}

}  // namespace mobgs
