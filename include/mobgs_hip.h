/*
 * mobgs_hip.h -- C ABI of libmobgs_hip.so, the MI355X (gfx950) Gaussian-splatting hot path for MoBGS.
 *
 * Boundary (level B3).  Each entry point replaces one stage of the third-party
 * gsplat==1.4.0 operator pipeline that the reference drives from
 *   /root/reference/gaussian_renderer/__init__.py:143-156   rasterization(...)        (and :163,201,236,255,274,
 *                                                                                       379,437,456,473,538)
 *   /root/reference/gaussian_renderer/__init__.py:190-199   fully_fused_projection(...) (and :411,422,513,524)
 * plus the reference's own per-Gaussian / per-pixel torch glue on the same path:
 *   /root/reference/gaussian_renderer/__init__.py:23-56     interpolate_cubic_hermite
 *   /root/reference/gaussian_renderer/__init__.py:93-125    time offset, rotation, colour feature build
 *   /root/reference/helper_model.py:19-28                   Sandwich colour decoder
 * and, either side of the path:
 *   /root/reference/scene/deformation.py, scene/hexplane.py deform_network (HexPlane + MLP heads)
 *   /root/reference/utils/loss_utils.py:233-239,351-381     L1 + SSIM
 *   /root/reference/main_utils.py:95-141                    normals from depth
 *   /root/reference/scene/gaussian_model.py:1044-1244,1352-1356,1480-1506   densification / optimiser surgery
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns every buffer (outputs and scratch); the library never allocates device memory; its only
 *     mutable state is the thread-local last-error string -- scheduling policy and testing switches travel with
 *     each call (MobgsTuning), so calls are re-entrant across streams, devices and threads;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no call synchronises, except
 *     mobgs_project_and_bin (one documented read-back);
 *   - return value: 0 on success, negative MOBGS_E_* on failure (mobgs_last_error() gives the text);
 *   - tensors are row-major contiguous float32 / int32 exactly as gsplat lays them out
 *     (means [N,3], quats [N,4] wxyz, scales [N,3], viewmats [C,4,4] world->camera, Ks [C,3,3],
 *      radii [C,N], means2d [C,N,2], depths [C,N], conics [C,N,3], colors [C,N,D] or [N,D],
 *      images [C,H,W,D], alphas [C,H,W]);
 *   - "tile" is a 16x16 pixel square (tile_size is fixed to 16 in this build, as every reference call site
 *     uses gsplat's default); tiles are numbered cam*tile_h*tile_w + ty*tile_w + tx.
 */
#ifndef MOBGS_HIP_H
#define MOBGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOBGS_OK 0
#define MOBGS_E_INVALID (-1)   /* bad argument (size, channel count, null pointer)            */
#define MOBGS_E_LAUNCH (-2)    /* hipLaunchKernel / hipMemsetAsync reported an error           */
#define MOBGS_E_UNSUPPORTED (-3)
#define MOBGS_E_CAPACITY (-4)   /* a caller-owned arena is too small; required sizes were reported         */

#define MOBGS_TILE 16
#define MOBGS_MAX_CHANNELS 32

/* Library identification.  Returns e.g. "mobgs_hip 0.2 gfx950". */
/* Per-call policy of the binning / compositing entry points.  The library keeps NO mutable state: what used to be
 * process-wide setters travels with each call (host pointer, may be NULL = all defaults; a negative field = default).
 *   heavy_tile_len     scheduling: a tile whose list has at least this many entries is composited by a whole
 *                      workgroup, one 8x8 quadrant per wave -- for at most an eighth of the tiles, or all of them
 *                      on grids of <= 1024 tiles (small images cannot fill the chip with one wave per tile).
 *                      Default 1024 (1 on grids of <= 1024 tiles); 0 = never.
 *   longest_list_hint  the longest per-tile list the caller expects (e.g. the previous frame's stats[2]).  Accepted
 *                      for compatibility: mobgs_isect_offsets now ranks through LDS on every grid of <= 8192 tiles
 *                      (it used to do so only with a hint >= 2048).  Default 0.
 *   quadrant_culling   testing aid, default 1: the compositors skip, per list entry, the 8x8 quadrants of the tile
 *                      the splat cannot reach (work that is predicated off at every pixel); 0 evaluates everything
 *                      -- results are identical.
 *   block_walk         forward compositor of the plain passes with <= 10 total channels, default 1: sixteen
 *                      independent 4x4-pixel workers per wave, each walking the entries that can reach ITS block
 *                      (raster.hip, "block-walk formulation"); 0 = the one-entry-at-a-time quadrant kernel.  Images,
 *                      alphas and last_ids are bit-identical either way.
 *   bwd_block_walk     experimental, default 0: 1 runs the same sixteen-worker formulation in the backward pass of the
 *                      7..10-channel plain passes (partial gradient records of the blocks combined in an LDS accumulator
 *                      under an integer claim).  Measured 9 % SLOWER than the quadrant kernel on the benchmark lists
 *                       -- kept as the A/B arm; gradients agree to summation order.
 *   geometry_per_camera  layout flag of mobgs_project_and_bin / _speculative (round 3), default 0: 1 = `means` is
 *                      [C,N,3] and `quats` [C,N,4] -- every camera sees its OWN positions / rotations of the N splats
 *                      (the K sub-frames of one blurry view: same Gaussians at K exposure times through K cameras,
 *                      projected, binned, sorted and composited as ONE C = K batch); scales / opacities stay [N,..].
 *   bwd_mfma           backward compositor of the passes with <= 10 total channels (round 4): the per-splat gradient
 *                      sums on the matrix pipe (raster_bwd_mfma.hip: the pair weights alpha*T and v_sigma are transposed
 *                      through LDS and summed over the pixels by v_mfma_f32_16x16x4_f32 against [colour cotangents |
 *                      pixel moments]; no per-entry wave reduction).  0 = OFF (the quadrant kernel with per-lane accumulators) -- NOT "default": a
 *                      zero-initialised struct selects the quadrant kernel on every grid, only a negative value defers to the library;
 *                      1 = one wave per tile, the four-wave team (one 8x8 quadrant per wave) for the schedule's heavy
 *                      tiles; 2 = the team for every tile.  Gradients agree to summation order (observed <= 2e-5 of
 *                      each tensor's maximum).  Default (-1): 1 on grids of <= 1024 tiles, else 0 (measured).
 *   gate_zero_cotangent  (round 5; was `reserved`) backward compositing passes, default 0 / -1 = off.  1: mobgs_raster_bwd
 *                      and mobgs_raster_class_bwd first ask, on the device, whether ANY element of v_render / v_alphas
 *                      is non-zero (one streaming kernel that stops at the first hit; NaN counts as non-zero).  If none
 *                      is, the compositing kernel returns at once, no gradient record is written, any_record stays 0
 *                      and mobgs_raster_bwd_reduce writes exact zeros without reading a slot: a loss term whose weight
 *                      is 0 (/root/reference/train.py:675 with arguments/stereo/seesaw.py lambda_flow_loss = 0) then
 *                      costs a probe instead of a backward pass.  No host synchronisation (HIP-graph safe).  Contract
 *                      in this mode: `any_record` points at TWO zeroed int32 words (the second is the gate; the usual
 *                      zero fill of grad_slots + the extra row behind them provides both).  One extra kernel launch per
 *                      call.  Ignored (treated as 0) with bwd_block_walk = 1.
 *   coherent_order     (round 5) mobgs_project_and_bin_fused, default 0 / -1 = unknown.  1: the caller states that the
 *                      splats are STORED in a spatially coherent order (neighbouring rows are neighbours in space, e.g.
 *                      rows along a Morton curve: mobgs_amd GaussianParams.spatial_sort_(), kept by TrainableGaussians
 *                      after every densification) -- the binning kernel then ranks through LDS on grids of up to 8192
 *                      tiles, as it does with an enum_order, without the order's indirection (measured at 300 k splats,
 *                      1352x1014: bin 49.6 -> 29.9 us; against an enum_order: scan 12.5 -> 7.1 us, slot reduction
 *                      38.9 -> 35.4 us).  A performance hint only: a wrong statement costs time, never correctness.
 *   static_rows        (round 6) mobgs_raster_bwd / mobgs_raster_class_bwd with 10 or 12 total channels, default 0 / -1 =
 *                      none.  S > 0: the caller states that the splats with (flat id % N) < S are MoBGS's STATIC set --
 *                      colour features cat(f_dc, 0.0 * f_t) (/root/reference/scene/gaussian_model.py:244-246; in
 *                      get_flow()'s 12-channel pass also a flow of exactly 0, gaussian_renderer/__init__.py:436-476) --
 *                      i.e. channels 6..8 (10 channels) / 6..10 (12 channels) of those rows are zero AND their gradient
 *                      is not wanted (the reference multiplies it by 0.0).  The backward compositor then runs a blend body
 *                      without those channels for such entries (one wave-uniform branch per list entry; 6 / 10 of ~50 VALU
 *                      per evaluated 8x8 quadrant) and v_colors[.., 6..] of those rows is written as 0.  Every other
 *                      output is BIT-identical (fma(0, v, acc) = acc).  The zeros are verified per staged entry (an entry
 *                      whose dead channels are not all +-0 takes the full body): a wrong statement about the DATA costs
 *                      nothing; a caller that does want d/d(colour 6..8) of such rows must leave this at 0.
 *   cover_slots        (round 6) mobgs_raster_bwd / mobgs_raster_bwd_decode, default 0 / -1 = off.  1: the caller did NOT
 *                      zero-fill grad_slots (108 MB at 1352x1014 / 300 k splats: a 15-us fill per render, 9 of them per
 *                      blurry view) -- the kernel writes the slot of EVERY entry of its lists, zeros where no pixel blended
 *                      the splat (entries behind every pixel's last blended one included), so mobgs_raster_bwd_reduce
 *                      sums exactly what it would have; stage 1 takes any_record = NULL then (the flag would live in unwritten
 *                      memory), stage 2 takes as its any_record the address of tile_offsets[n_tiles] -- the lists' total, 0
 *                      when a speculative binning call overflowed an arena and emptied the lists: without a host in the
 *                      loop (a HIP-graph replay) keep_scan and the slot ranges of such a frame must not be read.  Honoured only where the quadrant kernel runs ((mobgs_raster_path(D, 0, n_tiles,
 *                      tuning) & 3) == 0; other selections return MOBGS_E_UNSUPPORTED) and not together with
 *                      gate_zero_cotangent (a gated-off pass writes nothing).  Gradients are bit-identical. */
typedef struct MobgsTuning {
    int32_t heavy_tile_len;
    int32_t longest_list_hint;
    int32_t quadrant_culling;
    int32_t block_walk;
    int32_t bwd_block_walk;
    int32_t geometry_per_camera;
    int32_t bwd_mfma;
    int32_t gate_zero_cotangent;
    int32_t coherent_order;
    int32_t static_rows;
    int32_t cover_slots;
} MobgsTuning;

const char* mobgs_version(void);
/* Is any element of any of `n_arrays` float arrays (device pointers in the HOST array `arrays`, element counts in
 * `counts`) non-zero?  -> *live (device int32) = 1 if so, else 0; NaN counts as non-zero, -0.0 as zero.  Streams every
 * array once when all are zero, stops at the first hit otherwise.  The device-side form of "is this loss term's weight
 * zero" (MobgsTuning.gate_zero_cotangent uses the same kernel for one pass; mobgs_amd.gaussian_renderer reads *live on
 * the host to drop the whole backward graph of a view's get_flow() calls, /root/reference/train.py:570-579, :675). */
int mobgs_cotangent_probe(int n_arrays, const float* const* arrays, const size_t* counts, int32_t* live, void* stream);
/* Integer that changes whenever a signature, a struct layout or the format of a scratch buffer handed between entry
 * points changes (round 4 inserted `records` into mobgs_raster_bwd_reduce and changed the gradient-slot format without
 * one: a stale host extension would have passed shifted pointers).  Bindings compare it with the MOBGS_ABI_VERSION
 * they were built against and refuse to run on a mismatch (mobgs_amd/_lib.py, csrc/fastpath.cpp). */
#define MOBGS_ABI_VERSION 9
int mobgs_abi_version(void);
/* Text of the last error raised on the calling thread ("" if none). */
const char* mobgs_last_error(void);

/* Number of floats in one packed splat record / gradient slot for `channels` colour channels:
 * {x, y, conic_a, conic_b, conic_c, opacity, colour[channels]} rounded up to a multiple of 4. */
int mobgs_record_stride(int channels);

/* ---- K1: projection forward (replaces gsplat fully_fused_projection fwd) -------------------------------
 * Also emits tiles_per_gauss[C,N] (number of 16x16 tiles the 3-sigma box touches; 0 when culled), which
 * gsplat computes in isect_tiles' first pass.  Culled entries get radii=0 and zeros in the other outputs. */
int mobgs_project_fwd(int C, int N, const float* means, const float* quats, const float* scales,
                      const float* viewmats, const float* Ks, int width, int height, float eps2d,
                      float near_plane, float far_plane, float radius_clip, int32_t* radii,
                      float* means2d, float* depths, float* conics, int32_t* tiles_per_gauss,
                      void* stream);

/* ---- K2: projection backward (replaces gsplat fully_fused_projection bwd, incl. v_viewmats) ------------
 * v_viewmats_partial: scratch [ceil(C*N/256), 16] floats; v_viewmats [C,4,4] is fully written.
 * v_means/v_quats/v_scales are [N,3]/[N,4]/[N,3], summed over cameras, fully written by the call.
 * Any of v_means2d / v_depths / v_conics may be NULL (treated as zeros). */
size_t mobgs_project_bwd_scratch_floats(int C, int N);
int mobgs_project_bwd(int C, int N, const float* means, const float* quats, const float* scales,
                      const float* viewmats, const float* Ks, int width, int height, float eps2d,
                      const int32_t* radii, const float* conics, const float* v_means2d,
                      const float* v_depths, const float* v_conics, float* v_means, float* v_quats,
                      float* v_scales, float* v_viewmats, float* v_viewmats_partial, void* stream);
/* The same with geometry_per_camera (see MobgsTuning): means / v_means [C,N,3] and quats / v_quats [C,N,4] when the
 * flag is 1 (every camera's rows written once, nothing summed over cameras), v_scales [N,3] summed over cameras. */
int mobgs_project_bwd_ex(int C, int N, int geometry_per_camera, const float* means, const float* quats,
                         const float* scales, const float* viewmats, const float* Ks, int width, int height,
                         float eps2d, const int32_t* radii, const float* conics, const float* v_means2d,
                         const float* v_depths, const float* v_conics, float* v_means, float* v_quats,
                         float* v_scales, float* v_viewmats, float* v_viewmats_partial, void* stream);
/* Projection backward + prep backward of ONE camera in one launch (round 5; the backward half of
 * mobgs_prep_project_and_bin_fused): what mobgs_project_bwd would store as v_means / v_quats / v_scales stays in the
 * thread's registers and goes through the arithmetic of mobgs_prep_bwd (bit for bit) to the 13 leaf gradients.
 * x_means / x_quats / x_scales (each may be NULL): cotangents that reach the activated state directly, added first.
 * v_opacities / v_colors (may be NULL): as mobgs_prep_bwd.  accumulate: as mobgs_prep_bwd.  v_viewmats [4,4] is fully
 * written (scratch v_viewmats_partial: mobgs_project_bwd_scratch_floats(1, N) floats); v_viewmats = NULL: the pose needs no
 * gradient (the training loop never optimises it; eval.py's test-time pose optimisation does) -- the per-workgroup partial
 * rows and the reduction launch are skipped.  float32 leaves only. */
typedef struct MobgsLeafGrads {
    float *s_xyz, *s_scaling, *s_rotation, *s_opacity, *s_fdc, *s_ft, *d_control, *d_scaling, *d_rotation, *d_omega,
        *d_opacity, *d_fdc, *d_ft;
} MobgsLeafGrads;
int mobgs_project_prep_bwd_fused(int N, const float* means, const float* quats, const float* scales,
                                 const float* viewmats, const float* Ks, int width, int height, float eps2d,
                                 const int32_t* radii, const float* conics, const float* v_means2d,
                                 const float* v_depths, const float* v_conics, const float* x_means,
                                 const float* x_quats, const float* x_scales, float* v_viewmats,
                                 float* v_viewmats_partial, int Ns, int Nd, const float* times, const int64_t* d_ncp,
                                 const float* d_trbf, const float* opacities, const float* v_opacities,
                                 const float* v_colors, const MobgsLeafGrads* grads, int accumulate, void* stream);


/* ---- K3a: intersection offsets (replaces isect_tiles pass 1 + cumsum + isect_offset_encode) ------------
 * in : tiles_per_gauss [C*N] (bounding-box tile counts from mobgs_project_fwd), means2d, radii, conics,
 *      opacities ([C,N] when opac_per_camera else [N])
 * cull = 0: every bounding-box intersection is listed -- exactly upstream's lists.
 * cull = 1: a (tile, splat) pair is listed only if the splat can reach alpha >= 1/255 somewhere on the tile's
 *      pixel rectangle (min over the rectangle of sigma <= ln(255*opacity), conservative margin).  Pairs that
 *      fail are skipped by the compositor at all 256 pixels anyway, so every output pixel is bit-identical and every
 *      gradient keeps exactly its non-zero terms; lists, sort and gradient slots shrink (about 2x on anisotropic scenes).
 * out: cum_tiles [C*N+1] exclusive prefix sum of the BOX counts (cum_tiles[C*N] = I_box)
 *      keep_scan [mobgs_keep_scan_len(capacity)] exclusive prefix sum of the keep flags over the box
 *                intersections, stored in chunks of 2048 intersections, each preceded by one base word:
 *                chunk c = [base_c | local_0 .. local_2047], local_i = kept intersections before i inside the
 *                chunk, base_c = kept intersections in all earlier chunks.  The compact index of box intersection
 *                j (= its gradient slot in mobgs_raster_bwd) is
 *                    keep_scan[(j >> 11) * 2049] + keep_scan[(j >> 11) * 2049 + 1 + (j & 2047)],
 *                defined for 0 <= j <= min(I_box, capacity).
 *      tile_offsets [C*n_tiles+1] exclusive prefix sum of the per-tile list lengths
 *      tile_order [mobgs_tile_order_len(C*n_tiles)] (may be NULL) the schedule of the compositing kernels, 4
 *                slots per workgroup: tile ids by descending list length (1024 length classes; longest lists
 *                first, lists of similar length share a workgroup), -1 = unused slot.  Tiles whose list is at
 *                least MobgsTuning.heavy_tile_len long (capped, see there) appear as id | 1<<30 in
 *                the 4 slots of one workgroup, whose 4 waves then composite one 8x8 quadrant each.  A schedule
 *                only -- images do not depend on it; the compositing entry points accept NULL for raster order.
 *      stats int64[3] = {I_box, I_listed, longest per-tile list}; the caller reads them back to size the list
 *      buffers (the one host sync of the pipeline, as in gsplat).  If I_box > capacity the flags were
 *      truncated: call again with capacity >= I_box.
 * capacity_listed > 0 (speculative callers, see mobgs_isect_emit_sort_speculative): when I_box > capacity or
 *      I_listed > capacity_listed, tile_offsets is written as all zeros (every list empty) so that consumers
 *      already enqueued behind this call touch nothing; stats still hold the true counts.  <= 0: no such check.
 * scratch: mobgs_isect_scratch_bytes(C*N, C*n_tiles, capacity) bytes. */
size_t mobgs_isect_scratch_bytes(int n_gauss, int n_tiles, int capacity);
size_t mobgs_tile_order_len(int n_tiles);  /* int32 entries of tile_order */
size_t mobgs_keep_scan_len(int capacity); /* int32 entries of keep_scan for `capacity` box intersections */
int mobgs_isect_offsets(int C, int N, int tile_w, int tile_h, int width, int height, int cull, int capacity,
                        const int32_t* tiles_per_gauss, const float* means2d, const int32_t* radii,
                        const float* conics, const float* opacities, int opac_per_camera,
                        int32_t* cum_tiles, int32_t* keep_scan, int32_t* tile_offsets, int32_t* tile_order,
                        int64_t capacity_listed, int64_t* stats, void* scratch, const MobgsTuning* tuning,
                        void* stream);

/* ---- K3b/K4: emit + per-tile depth sort (replaces isect_tiles pass 2 + CUB DeviceRadixSort) ------------
 * Writes, per tile, its listed splats ordered by (float depth bits ascending, flat id ascending) -- the order a
 * stable LSD radix sort on gsplat's 64-bit key produces.
 * n_isects = stats[1], max_tile_len = stats[2] of mobgs_isect_offsets (lists longer than 4096 get a second launch
 * of the 1024-thread / 128-KiB-LDS sort variant over a compacted list of those tiles).
 * out: flatten_ids [n_isects] (cam*N+gaussian), isect_ids [n_isects] (gsplat's u64 key; may be NULL)
 * offsets_scratch: the SAME scratch buffer (and capacity) that was passed to mobgs_isect_offsets -- it holds the
 *   (flag, owner, tile, rank) of every bounding-box intersection (its dead per-tile counters are reused as
 *   scratch, so one call per mobgs_isect_offsets call); sort_keys [n_isects] u64 is scratch. */
int mobgs_isect_emit_sort(int C, int N, int tile_w, int tile_h, int capacity, int64_t n_isects,
                          int64_t max_tile_len, const float* depths, const int32_t* cum_tiles,
                          const int32_t* tile_offsets, const void* offsets_scratch, uint64_t* sort_keys,
                          int32_t* flatten_ids, uint64_t* isect_ids, void* stream);

/* The same stage enqueued BEFORE the counts are known on the host: the kernels read them from stats_dev (the
 * buffer mobgs_isect_offsets filled, called with the same capacity / capacity_listed) and do nothing when the
 * arena is too small.  max_tile_len_hint only selects the sort variant (any list length sorts correctly with
 * either; pass the previous frame's longest list).  sort_keys / flatten_ids / isect_ids hold capacity_listed
 * entries. */
int mobgs_isect_emit_sort_speculative(int C, int N, int tile_w, int tile_h, int capacity, int64_t capacity_listed,
                                      int64_t max_tile_len_hint, const float* depths, const int32_t* cum_tiles,
                                      const int32_t* tile_offsets, const int64_t* stats_dev,
                                      const void* offsets_scratch, uint64_t* sort_keys, int32_t* flatten_ids,
                                      uint64_t* isect_ids, void* stream);

/* ---- K1 + K3-K5 in one call: projection -> offsets (+ reach test) -> read-back -> emit -> per-tile sort ------
 * Same stages and buffers as mobgs_project_fwd + mobgs_isect_offsets + mobgs_isect_emit_sort, driven natively so
 * that no host gap separates the ~12 short kernels.  Arena: keep_scan [mobgs_keep_scan_len(capacity_box)], scratch
 * (mobgs_isect_scratch_bytes(C*N, C*n_tiles, capacity_box)), flatten_ids [capacity_listed], sort_keys
 * [capacity_listed], isect_ids [capacity_listed] or NULL.  stats_dev: 3 x int64 device scratch; stats_host: 3 x int64
 * HOST output {I_box, I_listed, longest list}.  Returns MOBGS_E_CAPACITY (stats_host valid, nothing written past
 * the arena) when I_box > capacity_box or I_listed > capacity_listed: grow and call again.  This function
 * synchronises `stream` once (the intersection-count read-back that upstream gsplat also performs). */
int mobgs_project_and_bin(int C, int N, const float* means, const float* quats, const float* scales,
                          const float* viewmats, const float* Ks, const float* opacities, int opac_per_camera,
                          int width, int height, float eps2d, float near_plane, float far_plane,
                          float radius_clip, int cull, int32_t* radii, float* means2d, float* depths,
                          float* conics, int32_t* tiles_per_gauss, int32_t* cum_tiles, int32_t* tile_offsets,
                          int32_t* tile_order, int64_t* stats_dev, int capacity_box, int32_t* keep_scan, void* scratch,
                          int64_t capacity_listed, int32_t* flatten_ids, uint64_t* sort_keys, uint64_t* isect_ids,
                          int64_t* stats_host, const MobgsTuning* tuning, void* stream);

/* mobgs_project_and_bin WITHOUT the host synchronisation: every stage is enqueued, the three counts are copied
 * asynchronously into stats_host_pinned (page-locked host memory, valid once the caller has waited on an event
 * recorded after this call) and the function returns.  When the arena turns out too small (counts exceed
 * capacity_box / capacity_listed) every per-tile list was written EMPTY: compositing kernels enqueued in the
 * meantime are harmless no-ops, and the caller redoes the binning with a larger arena (mobgs_isect_offsets +
 * mobgs_isect_emit_sort on the projection outputs, which do not depend on the arena).
 * stats_host_pinned holds FOUR words: {I_box, I_listed, longest list, sequence}.  With stats_seq != 0 and a pinned
 * buffer that is mapped into the device address space, the last binning kernel stores the three counts and then
 * stats_seq into word 3 (system-scope release): the caller clears word 3 beforehand and simply polls it -- no event,
 * hence no marker packet between the binning and the compositing kernels.  Return value 0 says so; 1 means the
 * counts travel by an asynchronous copy and the caller waits on an event recorded after this call, as before.
 * pack_records (optional, else NULL): [C*N, mobgs_record_stride(pack_channels + 1)] -- the projection kernel also
 * writes the compositor's packed records {means2d, conic, opacity, pack_colors, depth as the extra channel} of the
 * visible splats (what mobgs_pack_records / mobgs_raster_fwd would do in a launch of its own); hand them to
 * mobgs_raster_fwd with colors = NULL.  pack_colors: [C,N,ch] (colors_per_camera = 1) or [N,ch]. */
int mobgs_project_and_bin_speculative(int C, int N, const float* means, const float* quats, const float* scales,
                                      const float* viewmats, const float* Ks, const float* opacities,
                                      int opac_per_camera, int width, int height, float eps2d, float near_plane,
                                      float far_plane, float radius_clip, int cull, int32_t* radii, float* means2d,
                                      float* depths, float* conics, int32_t* tiles_per_gauss, int32_t* cum_tiles,
                                      int32_t* tile_offsets, int32_t* tile_order, int64_t* stats_dev,
                                      int capacity_box, int32_t* keep_scan, void* scratch, int64_t capacity_listed,
                                      int32_t* flatten_ids, uint64_t* sort_keys, uint64_t* isect_ids,
                                      int64_t max_tile_len_hint, int64_t* stats_host_pinned, int64_t stats_seq,
                                      const float* pack_colors, int colors_per_camera, int pack_channels,
                                      float* pack_records, const MobgsTuning* tuning, void* stream);

/* mobgs_project_and_bin_speculative with SINGLE-PASS lists (round 5).  The two-pass form ranks every kept intersection,
 * scans the per-tile counts, and only then scatters the sort keys (scan -> bin -> tile_scan -> emit -> sort, with a
 * second pass over the kept intersections in between).  Here the projection kernel also leaves one 48-byte "bin
 * record" per visible splat inside `scratch`, the binning kernel reads that one row per intersection and writes each
 * kept intersection's 64-bit key STRAIGHT into its tile's segment of a strided arena,
 *     seg_keys [C * n_tiles][8][seg_stride] u64   (mobgs_fused_seg_keys_len(C * n_tiles, seg_stride) entries; the 8
 *                                                  counter copies of the binning kernel each own a sub-segment),
 * and the per-tile sort reads the segments and writes flatten_ids / isect_ids PACKED at tile_offsets -- every output
 * is identical to the two-pass path's (same lists, same order, same keep_scan / cum_tiles / tile_order semantics).
 * seg_stride: capacity of a tile's list, 1 .. mobgs_fused_max_seg_stride(); the caller sizes it from the longest list
 * it expects (the previous frame's stats[2] plus slack).  A tile whose list is longer makes the call behave like an
 * arena overflow: every list is written EMPTY, stats still hold the true counts (stats[2] > seg_stride tells), and
 * the caller redoes the binning with the two-pass entry points.  Requires N > 0, capacity_box >= 4 C N + 2 and a
 * 128-byte aligned scratch.  Everything else as mobgs_project_and_bin_speculative (including the return value).
 * enum_order (optional, else NULL): [C*N] int32, a PERMUTATION of the flat splat ids 0 .. C*N-1 -- the order in which the
 *   splats' bounding-box intersections are enumerated (cum_tiles / keep_scan / gradient-slot numbering).  Any permutation
 *   gives the same lists, images and gradients (each splat's intersections stay consecutive and in the same order); a
 *   SPATIALLY COHERENT one (e.g. a Morton order of the positions, recomputed now and then by the caller) makes
 *   neighbouring intersections hit neighbouring tiles, so the binning kernel ranks them in LDS and issues an order of
 *   magnitude fewer returning atomics (48 -> 27 us at 300 k splats).  With a permutation cum_tiles[g + 1] is no longer
 *   the END of splat g's intersections: pass tiles_per_gauss to mobgs_raster_bwd_reduce / mobgs_raster_layers_bwd. */
size_t mobgs_fused_seg_keys_len(int n_tiles, int seg_stride);
int mobgs_fused_max_seg_stride(void);
int mobgs_project_and_bin_fused(int C, int N, const float* means, const float* quats, const float* scales,
                                const float* viewmats, const float* Ks, const float* opacities, int opac_per_camera,
                                int width, int height, float eps2d, float near_plane, float far_plane,
                                float radius_clip, int cull, int32_t* radii, float* means2d, float* depths,
                                float* conics, int32_t* tiles_per_gauss, int32_t* cum_tiles, int32_t* tile_offsets,
                                int32_t* tile_order, int64_t* stats_dev, int capacity_box, int32_t* keep_scan,
                                void* scratch, int64_t capacity_listed, int32_t* flatten_ids, uint64_t* seg_keys,
                                int seg_stride, const int32_t* enum_order, uint64_t* isect_ids, int64_t max_tile_len_hint,
                                int64_t* stats_host_pinned, int64_t stats_seq, const float* pack_colors,
                                int colors_per_camera, int pack_channels, float* pack_records,
                                const MobgsTuning* tuning, void* stream);
/* The same call with the per-splat state built INSIDE the projection kernel (round 5): what
 * mobgs_prep_fwd computes from the raw parameters of the two sets -- /root/reference/gaussian_renderer/__init__.py:93-125,
 * :181-185 (spline position, rotation + t * omega, exp / sigmoid activations, colour features) -- is evaluated by the
 * thread that projects the splat, bit for bit the arithmetic of mobgs_prep_fwd, so the activated state is not written
 * by one launch and read back by the next.  One camera (C = 1), float attributes, N = prep->Ns + prep->Nd.
 * `means` [N,3], `quats` [N,4], `scales` [N,3] and `opacities` [N] are OUTPUTS here (the backward pass and the caller's
 * result dict need them); the 9 colour features only exist inside `pack_records` (stride mobgs_record_stride(10): the
 * depth is the tenth channel), which is required.  All pointers in the struct are device pointers. */
typedef struct MobgsPrepInputs {
    int32_t Ns, Nd;
    const float* times;      /* [2]: {t_feat, t_curve}, as mobgs_prep_fwd */
    const float *s_xyz, *s_scaling, *s_rotation, *s_opacity, *s_fdc, *s_ft;
    const float* d_control;
    const int64_t* d_ncp;
    const float *d_scaling, *d_rotation, *d_omega, *d_opacity, *d_fdc, *d_ft, *d_trbf;
} MobgsPrepInputs;
int mobgs_prep_project_and_bin_fused(const MobgsPrepInputs* prep, float* means, float* quats, float* scales,
                                     const float* viewmats, const float* Ks, float* opacities, int width, int height,
                                     float eps2d, float near_plane, float far_plane, float radius_clip, int cull,
                                     int32_t* radii, float* means2d, float* depths, float* conics,
                                     int32_t* tiles_per_gauss, int32_t* cum_tiles, int32_t* tile_offsets,
                                     int32_t* tile_order, int64_t* stats_dev, int capacity_box, int32_t* keep_scan,
                                     void* scratch, int64_t capacity_listed, int32_t* flatten_ids, uint64_t* seg_keys,
                                     int seg_stride, const int32_t* enum_order, uint64_t* isect_ids,
                                     int64_t max_tile_len_hint, int64_t* stats_host_pinned, int64_t stats_seq,
                                     float* pack_records, const MobgsTuning* tuning, void* stream);


/* ---- K6: rasterise forward (replaces gsplat rasterize_to_pixels fwd) -----------------------------------
 * colors   : [C,N,channels] (colors_per_camera=1) or [N,channels] (0); NULL: `records` are already packed (by
 *            mobgs_project_and_bin_speculative or mobgs_pack_records) and only `extra != NULL` is looked at
 * opacities: [C,N] (opac_per_camera=1) or [N] (0)
 * extra    : optional [C,N] channel appended after `channels` (gsplat's "+D"/"+ED" depth channel), or NULL
 * backgrounds: [C, channels(+1 if extra)] or NULL
 * records  : scratch/out [C*N, mobgs_record_stride(D)] packed splat records (kept for backward; an OPAQUE format of
 *            the kernels: {x, y, A, B, C, L, colours..} with conic and opacity in exponent form,
 *            opacity exp(-sigma) = exp2(A dx^2 + C dy^2 + B dx dy + L) -- csrc/common.h write_splat_record), D = total
 * out: render [C,H,W,D], alphas [C,H,W], last_ids [C,H,W] (index into flatten_ids of the last blended splat)
 * isect_reach (optional out, else NULL): [I_listed] bytes -- per list entry, the 8x8 quadrants of its tile the splat
 *   can reach (what the kernel computes anyway to skip the others); hand it to mobgs_raster_bwd for the same lists and
 *   the backward pass does not recompute it */
int mobgs_raster_fwd(int C, int N, int channels, int width, int height, const float* means2d,
                     const float* conics, const float* colors, int colors_per_camera,
                     const float* opacities, int opac_per_camera, const float* extra,
                     const float* backgrounds, const int32_t* radii, const int32_t* tile_offsets,
                     const int32_t* tile_order, const int32_t* flatten_ids, float* records, float* render,
                     float* alphas, int32_t* last_ids, uint8_t* isect_reach, const MobgsTuning* tuning,
                     void* stream);

/* mobgs_raster_fwd + the Sandwich decoder as the compositor's EPILOGUE (round 5): for the 9-feature + depth pass of
 * render() (channels = 9, extra != NULL, MobgsTuning.block_walk on) the kernel also writes what mobgs_decoder_fwd_many
 * would compute from `render` / `alphas` -- rgb [C,3,H,W] and the expected depth [C,H,W] -- while the pixel's features are
 * still in registers: no decoder launch and no re-read of the feature image (which is written all the same: the backward
 * passes read it).  Pinhole rays only: ray_intr [fx, fy, cx, cy] and ray_c2w (the first 12 entries of the row-major
 * camera-to-world matrix) per camera, intr_stride / c2w_stride floats apart (0 = shared); w1 [6,12], w2 [3,6] as
 * /root/reference/helper_model.py:19-28.  Bit-identical to the two separate calls.  Other channel counts / the quadrant
 * kernel: MOBGS_E_UNSUPPORTED (call mobgs_raster_fwd and mobgs_decoder_fwd_many). */
int mobgs_raster_fwd_decode(int C, int N, int channels, int width, int height, const float* means2d,
                            const float* conics, const float* colors, int colors_per_camera, const float* opacities,
                            int opac_per_camera, const float* extra, const float* backgrounds, const int32_t* radii,
                            const int32_t* tile_offsets, const int32_t* tile_order, const int32_t* flatten_ids,
                            float* records, float* render, float* alphas, int32_t* last_ids, uint8_t* isect_reach,
                            const float* ray_intr, int intr_stride, const float* ray_c2w, int c2w_stride, const float* w1,
                            const float* w2, float* rgb, float* depth, const MobgsTuning* tuning, void* stream);

/* ---- K7: rasterise backward (replaces gsplat rasterize_to_pixels bwd) ----------------------------------
 * Deterministic two-stage gradient reduction, no floating-point atomics:
 *   stage 1 (mobgs_raster_bwd) walks every tile back to front and writes ONE gradient record
 *     {A, B, Sxx, Sxy, Syy, S, v_colour[..]} per (tile, splat) intersection -- the geometry and opacity terms as raw
 *     sums over the tile's pixels, with d = mean2d - pixel centre and v_sigma the cotangent of the exponent:
 *     A = sum v_sigma dx, B = sum v_sigma dy, Sxx = sum v_sigma dx^2, Sxy = sum v_sigma dx dy, Syy = sum v_sigma dy^2,
 *     S = sum v_sigma -- into
 *     grad_slots [I_listed, stride]; slot = compact index (see keep_scan) of cum_tiles[flat id] + position of the tile inside the
 *     splat's tile rectangle].  grad_slots must be zero-filled by the caller (intersections that no pixel blended stay 0).
 *   stage 2 (mobgs_raster_bwd_reduce) sums each splat's contiguous slots and applies the splat's conic (a, b, c) of
 *     `records` (the packed records stage 1 read) once: v_means2d = (a A + b B, b A + c B), v_conics = (Sxx / 2, Sxy,
 *     Syy / 2), v_opacity = -S / opacity (a pair's v_sigma is -opacity vis v_alpha) -- conic and opacity are constant
 *     per splat, so this equals summing gsplat's per-pixel terms -- into the dense
 *     gradients v_means2d [C,N,2], v_conics [C,N,3], v_opacities [C,N], v_colors [C,N,channels], v_extra [C,N] (NULL
 *     when there was no extra channel); all fully written.
 *   any_record (optional, both stages the same zero-initialised int32; NULL = off): stage 1 sets it when it writes
 *     its first record, stage 2 reads no slot when it is still 0 -- a pass whose cotangents are all exactly zero (a
 *     loss term with weight 0, e.g. lambda_flow_loss = 0 in arguments/stereo/seesaw.py) then costs no reduction. */
int mobgs_raster_bwd(int C, int N, int channels, int has_extra, int width, int height,
                     const float* records, const float* backgrounds, const int32_t* radii,
                     const float* means2d, const int32_t* cum_tiles, const int32_t* keep_scan,
                     const int32_t* tile_offsets, const int32_t* tile_order, const int32_t* flatten_ids,
                     const float* render_alphas, const int32_t* last_ids, const float* v_render,
                     const float* v_alphas, float* grad_slots, const uint8_t* isect_reach,
                     int32_t* any_record, const MobgsTuning* tuning, void* stream);
/* Stage 1 with the Sandwich decoder's BACKWARD pass as its prologue (round 6; the counterpart of mobgs_raster_fwd_decode;
 * replaces, for the lean render() of /root/reference/gaussian_renderer/__init__.py:201-227, the chain
 * [helper_model.py:19-28 backward -> gsplat rasterize_to_pixels backward]): 9 feature channels + the depth channel,
 * pinhole rays.  The kernel reads the decoder's inputs -- `render` [C,H,W,10] (the composited image mobgs_raster_fwd[_decode]
 * wrote), `render_alphas`, the cotangents v_rgb [C,3,H,W] of the decoded colour and v_depth [C,H,W] (NULL = none) of the
 * expected depth -- and forms the cotangent of the composited image in registers (bit for bit what mobgs_decoder_bwd
 * would have written to v_feat_hw / v_alphas; v_alphas here = an ADDITIONAL cotangent of the alpha output, NULL = none).
 * The weight and pose gradients leave the kernel as ONE row of 102 sums per tile in w_partial (scratch of
 * mobgs_raster_bwd_decode_scratch_floats(C, width, height) floats, contents irrelevant on entry: the rows, then chunk sums
 * and a ticket word); mobgs_raster_bwd_decode_finish -- any time later on the same stream -- adds them in a fixed order
 * (deterministic) into g_w1 [6,12], g_w2 [3,6] (accumulate_wgrad != 0: added to what is there; sums over all images) and
 * g_c2w [C, g_c2w_floats = 12 | 16] (NULL = not wanted; per image, fully written).  Only where the quadrant kernel is the
 * selection ((mobgs_raster_path(10, 0, n_tiles, tuning) & 3) == 0), else MOBGS_E_UNSUPPORTED: callers then run
 * mobgs_decoder_bwd + mobgs_raster_bwd.  Everything else as mobgs_raster_bwd (channels = 9, has_extra = 1). */
int mobgs_raster_bwd_decode(int C, int N, int width, int height, const float* records, const float* backgrounds,
                            const int32_t* radii, const int32_t* cum_tiles, const int32_t* keep_scan,
                            const int32_t* tile_offsets, const int32_t* tile_order, const int32_t* flatten_ids,
                            const float* render, const float* render_alphas, const int32_t* last_ids, const float* v_rgb,
                            const float* v_depth, const float* v_alphas, const float* ray_intr, int intr_stride,
                            const float* ray_c2w, int c2w_stride, const float* w1, const float* w2, float* grad_slots,
                            const uint8_t* isect_reach, int32_t* any_record, float* w_partial, const MobgsTuning* tuning,
                            void* stream);
int mobgs_raster_bwd_decode_finish(int C, int width, int height, float* w_partial, int c2w_stride, float* g_w1,
                                   float* g_w2, float* g_c2w, int g_c2w_floats, int accumulate_wgrad, void* stream);
/* mobgs_raster_bwd_reduce (channels = 9, has_extra = 1) and mobgs_raster_bwd_decode_finish in ONE launch: the weight / pose
 * sums run as extra leading workgroups of the slot reduction (no launch of their own).  Same results as the two calls
 * (weight / pose gradients to summation order). */
int mobgs_raster_bwd_reduce_decode(int C, int N, const float* records, const int32_t* cum_tiles, const int32_t* keep_scan,
                                   const float* grad_slots, const int32_t* any_record, float* v_means2d, float* v_conics,
                                   float* v_opacities, float* v_colors, float* v_extra, const int32_t* tiles_per_gauss,
                                   int width, int height, const float* w_partial, int c2w_stride, float* g_w1, float* g_w2,
                                   float* g_c2w, int g_c2w_floats, int accumulate_wgrad, void* stream);
size_t mobgs_raster_bwd_decode_scratch_floats(int C, int width, int height);
/* tiles_per_gauss (optional, else NULL): [C*N] -- splat g's intersections are then [cum_tiles[g], cum_tiles[g] +
 *     tiles_per_gauss[g]) instead of [cum_tiles[g], cum_tiles[g + 1]): REQUIRED when the lists were built with an enum_order
 *     (mobgs_project_and_bin_fused), equivalent otherwise. */
int mobgs_raster_bwd_reduce(int C, int N, int channels, int has_extra, const float* records,
                            const int32_t* cum_tiles, const int32_t* keep_scan, const float* grad_slots,
                            const int32_t* any_record, float* v_means2d, float* v_conics, float* v_opacities,
                            float* v_colors, float* v_extra, const int32_t* tiles_per_gauss, void* stream);

/* Packs the compositor's per-splat inputs into records [C*N, mobgs_record_stride(D)] (the first step of
 * mobgs_raster_fwd, exposed for mobgs_raster_layers_fwd). */
int mobgs_pack_records(int C, int N, int channels, const float* means2d, const float* conics, const float* colors,
                       int colors_per_camera, const float* opacities, int opac_per_camera, const float* extra,
                       const int32_t* radii, float* records, void* stream);

/* ---- K6'/K7': layered compositing (train-mode render(): combined + static-only + dynamic-only in one pass) ----
 * Replaces the 5 rasterization() calls of /root/reference/gaussian_renderer/__init__.py:143-176,201-214,236-268
 * that share one camera: splats with (flat id % N) < Ns are "static", the others "dynamic".
 * layer_mask (non-zero): bit 0 = all, bit 1 = static-only, bit 2 = dynamic-only.  10 total channels only
 * (9 features + depth).  Per-layer tensors are passed as HOST arrays of 3 device pointers (index = layer;
 * entries of layers that are not requested may be NULL): render [C,H,W,10], alphas [C,H,W], last_ids [C,H,W].
 * Backward: v_render3 / v_alphas3 entries may be NULL (zero cotangent); grad_slots [I_listed, 2, 16] and
 * grad_xy0 [I_listed, 2, 2] are zero-filled scratch (one record per HALF tile and splat); the dense gradients
 * are fully written.  v_means2d is the total over the layers, v_means2d_layer0 the share of the combined render
 * alone (what the reference's `viewspace_points.grad` holds, gaussian_renderer/__init__.py:218-223). */
int mobgs_raster_layers_fwd(int C, int N, int Ns, int layer_mask, int channels_total, int width, int height,
                            const float* records, const float* backgrounds, const int32_t* tile_offsets,
                            const int32_t* tile_order, const int32_t* flatten_ids, float* const* render3_host,
                            float* const* alphas3_host,
                            int32_t* const* last_ids3_host, void* stream);
int mobgs_raster_layers_bwd(int C, int N, int Ns, int layer_mask, int channels, int has_extra, int width,
                            int height, const float* records, const float* backgrounds, const int32_t* radii,
                            const int32_t* cum_tiles, const int32_t* keep_scan, const int32_t* tile_offsets,
                            const int32_t* tile_order, const int32_t* flatten_ids,
                            const float* const* render_alphas3_host,
                            const int32_t* const* last_ids3_host, const float* const* v_render3_host,
                            const float* const* v_alphas3_host, float* grad_slots, float* grad_xy0,
                            float* v_means2d_layer0, float* v_means2d, float* v_conics, float* v_opacities,
                            float* v_colors, float* v_extra, const int32_t* tiles_per_gauss, void* stream);

/* ---- K13: densification ---------------------------------------------------------------
 * The per-splat table (14 optimiser groups + their Adam moments + 5 statistics arrays) of
 * /root/reference/scene/gaussian_model.py:598-617 is resized by clone / split / prune
 * (:1044-1155, :1207-1244, :1480-1506).  The reference does it with ~50 torch index/cat/repeat calls per
 * operation; here a resize is: masks (mobgs_densify_select) -> row list (mobgs_mask_indices + list arithmetic) ->
 * ONE mobgs_rows_gather launch that moves every field into a second buffer set -> mobgs_split_children.
 *
 * mobgs_densify_stats: the per-iteration statistics of helper_train.py:263-264 + gaussian_model.py:1352-1356,
 *   for rows with visible[i] != 0 (visible == NULL: radii[i] > 0): max_radii2D[i] = max(., radii[i]) (skipped when
 *   radii or max_radii2D is NULL), xyz_gradient_accum[i] += |viewspace_grad[i, 0:2]|, denom[i] += 1.
 *   viewspace_grad is [n, grad_stride] (2 for means2d.grad, 3 for the reference's screenspace tensor).
 * mobgs_densify_select: g = accum/denom (NaN -> 0; rows >= n_grads count as 0), big = max_k exp(scaling[i,k]) >
 *   size_threshold (= percent_dense * scene_extent): clone_sel = |g| >= thr && !big, split_sel = g >= thr && big.
 * mobgs_mask_indices: indices[0..count) = ascending rows with (mask[i] != 0) == (want != 0); count is a device int.
 * mobgs_rows_gather: for every field f (host arrays of n_fields device pointers / row sizes in bytes / flags) and
 *   output row r: dst_f[dst_offset + r] = src_f[index[r]] if index[r] >= 0; a negative index marks a NEW copy of
 *   row -(index[r]+1): fields with zero_new[f] != 0 (Adam moments) get zeros there, the others the copy.
 * mobgs_split_children: rows [first_row, first_row + n_children) hold copies of their parents;
 *   xyz += R(rotation) * sample, scaling = log(exp(scaling) / (0.8 * n_split))  (gaussian_model.py:1218-1221). */
int mobgs_densify_stats(int n, const float* viewspace_grad, int grad_stride, const uint8_t* visible,
                        const int32_t* radii, float* xyz_gradient_accum, float* denom, float* max_radii2D,
                        void* stream);
int mobgs_densify_select(int n, int n_grads, const float* xyz_gradient_accum, const float* denom,
                         const float* scaling, float grad_threshold, float size_threshold, uint8_t* clone_sel,
                         uint8_t* split_sel, void* stream);
int mobgs_mask_indices(int n, const uint8_t* mask, int want, int32_t* indices, int32_t* count, void* stream);
int mobgs_rows_gather(int n_fields, const void* const* src_host, void* const* dst_host,
                      const int32_t* row_bytes_host, const int32_t* zero_new_host, const int32_t* index, int n_out,
                      int dst_offset, void* stream);
int mobgs_split_children(int n_children, int first_row, int n_split, const float* samples, const float* rotation,
                         float* xyz, float* scaling, void* stream);

/* ---- K14: normals from a depth map (replaces /root/reference/main_utils.py:95-141 get_normals, train.py:590) ----
 * z [H,W] -> normals [3,H,W]: every pixel is back-projected with its view direction
 *   y = (i + pixel_offset - cy) / fy,  x = (j + pixel_offset - cx - y * skew) / fx,  point = (x, y, 1) * z,
 * n = normalize(cross(right - left, top - bottom), eps 1e-12) on the interior, zeros on the 1-pixel border.
 * pixel_offset = 0.5 for dycheck cameras with use_center (the default).  Backward: v_z [H,W] fully written. */
int mobgs_normals_fwd(int H, int W, float fx, float fy, float cx, float cy, float skew, float pixel_offset,
                      const float* z, float* normals, void* stream);
int mobgs_normals_bwd(int H, int W, float fx, float fy, float cx, float cy, float skew, float pixel_offset,
                      const float* z, const float* v_normals, float* v_z, void* stream);

/* ---- K6''/K7'': class-restricted passes of the single-set compositor -----------------------------------------
 * Replaces the static-only / dynamic-only rasterization() calls of
 * /root/reference/gaussian_renderer/__init__.py:201-214 (dynamic), :236-250 (static) and their ones-colour alpha
 * passes :163-177, :255-269, for splat sets that are the two halves of one projected set.
 * The static-only (class_sel = 1: flat id % N < Ns) or dynamic-only (class_sel = 2) "RGB+D" render over the lists
 * of the WHOLE set: entries of the other class are dropped as each 64-entry batch is staged.  Every splat belongs
 * to exactly one class, so the two passes together blend each (tile, splat) pair once and their backward passes
 * write disjoint records of ONE grad_slots buffer (zero-filled [I_listed, stride]), reduced by one
 * mobgs_raster_bwd_reduce.  channels_total = 10 (9 features + depth), or 1 (one colour channel, no extra: the
 * dynamic-only coverage of get_flow(), /root/reference/gaussian_renderer/__init__.py:477-490, where only the alpha
 * output matters); records from mobgs_pack_records; last_ids index the whole set's lists; any_record as in
 * mobgs_raster_bwd. */
int mobgs_raster_class_fwd(int C, int N, int Ns, int class_sel, int channels_total, int width, int height,
                           const float* records, const float* backgrounds, const int32_t* tile_offsets,
                           const int32_t* tile_order, const int32_t* flatten_ids, float* render, float* alphas,
                           int32_t* last_ids, uint8_t* isect_reach, const MobgsTuning* tuning, void* stream);
int mobgs_raster_class_bwd(int C, int N, int Ns, int class_sel, int channels_total, int width, int height,
                           const float* records, const float* backgrounds, const int32_t* radii,
                           const int32_t* cum_tiles, const int32_t* keep_scan, const int32_t* tile_offsets,
                           const int32_t* tile_order, const int32_t* flatten_ids, const float* render_alphas,
                           const int32_t* last_ids, const float* v_render, const float* v_alphas, float* grad_slots,
                           const uint8_t* isect_reach, int32_t* any_record, const MobgsTuning* tuning, void* stream);

/* 1 if raster kernels are compiled for `total_channels` (colour channels + optional extra channel). */
int mobgs_raster_channels_supported(int total_channels);

/* Which compositing kernels a pass of `total_channels` over a grid of `n_tiles` tiles (all cameras) takes under
 * `tuning` -- the decision functions the launchers themselves use, exported so that a test can ASSERT which kernels a
 * comparison exercised (round-5 review: every reference-generated fixture ran the small-grid selection, the benchmark
 * the other one).  class_filter: 1 for mobgs_raster_class_fwd/bwd.  Returns a bit field:
 *   bits 0-1  backward kernel: 0 = quadrant kernel raster_bwd_kernel (per-lane accumulators + wave reduction),
 *             1 = matrix-pipe backward, one wave per tile + four-wave team for heavy tiles, 2 = team for every tile,
 *             3 = block-walk backward (MobgsTuning.bwd_block_walk)
 *   bit 2     forward: 1 = block-walk kernel raster_fwd_blocks_kernel (the decoder epilogue exists only there)
 *   bit 3     the tile schedule may mark tiles heavy (four waves per tile) -- whether any IS heavy depends on the list
 *             lengths: count the entries of tile_order with bit 30 set
 *   bits 8-.. the list length from which a tile is scheduled heavy (0 = never)
 * Not a launch: no device work, no error state. */
int mobgs_raster_path(int total_channels, int class_filter, int n_tiles, const MobgsTuning* tuning);

/* ---- K8: per-splat state build (replaces the reference's per-Gaussian torch glue) -----------------------
 * /root/reference/gaussian_renderer/__init__.py:23-56 (interpolate_cubic_hermite), :93-125 (time offset,
 * rotation, colour features), :181-185 (cat static|dynamic); scene/gaussian_model.py:209-254 (activations).
 * Rows [0,Ns) of every output are the static splats, [Ns,Ns+Nd) the dynamic ones.
 * times (device, 2 floats): {t_feat = time + delta/max_time, t_curve = clamp(t_feat,0,1)}.
 * static leaves : xyz [Ns,3], scaling [Ns,3] (log), rotation [Ns,4], opacity [Ns] (logit), f_dc [Ns,6], f_t [Ns,3]
 * dynamic leaves: control [Nd,12,3] (x100 units), ncp [Nd] int64 active knots (4..12), scaling, rotation,
 *                 omega [Nd,4], opacity, f_dc, f_t, trbf [Nd]
 * out: means [N,3], quats [N,4] (un-normalised; the projection normalises), scales [N,3], opacities [N],
 *      colors [N,9]. */
int mobgs_prep_fwd(int Ns, int Nd, const float* times, const float* s_xyz, const float* s_scaling,
                   const float* s_rotation, const float* s_opacity, const float* s_fdc, const float* s_ft,
                   const float* d_control, const int64_t* d_ncp, const float* d_scaling,
                   const float* d_rotation, const float* d_omega, const float* d_opacity, const float* d_fdc,
                   const float* d_ft, const float* d_trbf, float* means, float* quats, float* scales,
                   float* opacities, float* colors, void* stream);
/* Backward of mobgs_prep_fwd.  Cotangent pointers may be NULL (zeros).  Every gradient buffer is fully
 * written.  trbf and the times get no gradient (the reference detaches the time offset, :102). */
int mobgs_prep_bwd(int Ns, int Nd, const float* times, const int64_t* d_ncp, const float* d_trbf,
                   const float* scales, const float* opacities, const float* v_means, const float* v_quats,
                   const float* v_scales, const float* v_opacities, const float* v_colors, float* g_s_xyz,
                   float* g_s_scaling, float* g_s_rotation, float* g_s_opacity, float* g_s_fdc, float* g_s_ft,
                   float* g_d_control, float* g_d_scaling, float* g_d_rotation, float* g_d_omega,
                   float* g_d_opacity, float* g_d_fdc, float* g_d_ft, int accumulate, void* stream);
/* The same two entry points for Gaussian sets whose ATTRIBUTES (scaling, rotation, omega, opacity, f_dc, f_t) are
 * stored as IEEE binary16 (BASELINE config #5; no such mode exists in the reference, whose GaussianModel keeps
 * fp32 nn.Parameters, scene/gaussian_model.py:1044-1090): the kernels read the halves from HBM and widen them in
 * registers, arithmetic and outputs stay fp32.  xyz / control / trbf stay fp32.  mobgs_prep_bwd_f16 writes the
 * attribute gradients as binary16 (what a half leaf's .grad must be); accumulation across renders in fp32 buffers
 * goes through mobgs_prep_bwd, which does not read the attributes. */
int mobgs_prep_fwd_f16(int Ns, int Nd, const float* times, const float* s_xyz, const uint16_t* s_scaling,
                       const uint16_t* s_rotation, const uint16_t* s_opacity, const uint16_t* s_fdc,
                       const uint16_t* s_ft, const float* d_control, const int64_t* d_ncp,
                       const uint16_t* d_scaling, const uint16_t* d_rotation, const uint16_t* d_omega,
                       const uint16_t* d_opacity, const uint16_t* d_fdc, const uint16_t* d_ft, const float* d_trbf,
                       float* means, float* quats, float* scales, float* opacities, float* colors, void* stream);
int mobgs_prep_bwd_f16(int Ns, int Nd, const float* times, const int64_t* d_ncp, const float* d_trbf,
                       const float* scales, const float* opacities, const float* v_means, const float* v_quats,
                       const float* v_scales, const float* v_opacities, const float* v_colors, float* g_s_xyz,
                       uint16_t* g_s_scaling, uint16_t* g_s_rotation, uint16_t* g_s_opacity, uint16_t* g_s_fdc,
                       uint16_t* g_s_ft, float* g_d_control, uint16_t* g_d_scaling, uint16_t* g_d_rotation,
                       uint16_t* g_d_omega, uint16_t* g_d_opacity, uint16_t* g_d_fdc, uint16_t* g_d_ft,
                       int accumulate, void* stream);

/* The four entry points above for K time instants in ONE launch (the K sub-frames of a blurry view, train.py:502-518: the
 * same Gaussians at K exposure times): times [K,2]; means [K,N,3], quats [K,N,4], colors [K,N,9] (row block k = instant
 * k); scales [N,3] and opacities [N] do not depend on time and exist once.  Backward: v_means / v_quats / v_colors
 * [K,...], v_scales / v_opacities once; the leaf gradients are the sums over the instants, accumulated in instant order
 * (the result of K single-instant calls in a row, bit for bit).  K = 1 is the single-instant call. */
int mobgs_prep_fwd_many(int K, int Ns, int Nd, const float* times, const float* s_xyz, const float* s_scaling,
                        const float* s_rotation, const float* s_opacity, const float* s_fdc, const float* s_ft,
                        const float* d_control, const int64_t* d_ncp, const float* d_scaling,
                        const float* d_rotation, const float* d_omega, const float* d_opacity, const float* d_fdc,
                        const float* d_ft, const float* d_trbf, float* means, float* quats, float* scales,
                        float* opacities, float* colors, void* stream);
int mobgs_prep_bwd_many(int K, int Ns, int Nd, const float* times, const int64_t* d_ncp, const float* d_trbf,
                        const float* scales, const float* opacities, const float* v_means, const float* v_quats,
                        const float* v_scales, const float* v_opacities, const float* v_colors, float* g_s_xyz,
                        float* g_s_scaling, float* g_s_rotation, float* g_s_opacity, float* g_s_fdc, float* g_s_ft,
                        float* g_d_control, float* g_d_scaling, float* g_d_rotation, float* g_d_omega,
                        float* g_d_opacity, float* g_d_fdc, float* g_d_ft, int accumulate, void* stream);
int mobgs_prep_fwd_many_f16(int K, int Ns, int Nd, const float* times, const float* s_xyz, const uint16_t* s_scaling,
                            const uint16_t* s_rotation, const uint16_t* s_opacity, const uint16_t* s_fdc,
                            const uint16_t* s_ft, const float* d_control, const int64_t* d_ncp,
                            const uint16_t* d_scaling, const uint16_t* d_rotation, const uint16_t* d_omega,
                            const uint16_t* d_opacity, const uint16_t* d_fdc, const uint16_t* d_ft,
                            const float* d_trbf, float* means, float* quats, float* scales, float* opacities,
                            float* colors, void* stream);
int mobgs_prep_bwd_many_f16(int K, int Ns, int Nd, const float* times, const int64_t* d_ncp, const float* d_trbf,
                            const float* scales, const float* opacities, const float* v_means, const float* v_quats,
                            const float* v_scales, const float* v_opacities, const float* v_colors, float* g_s_xyz,
                            uint16_t* g_s_scaling, uint16_t* g_s_rotation, uint16_t* g_s_opacity, uint16_t* g_s_fdc,
                            uint16_t* g_s_ft, float* g_d_control, uint16_t* g_d_scaling, uint16_t* g_d_rotation,
                            uint16_t* g_d_omega, uint16_t* g_d_opacity, uint16_t* g_d_fdc, uint16_t* g_d_ft,
                            int accumulate, void* stream);

/* ---- K9: colour decoder + expected-depth normalisation (replaces Sandwich.forward + gsplat's "ED" step) ---
 * /root/reference/helper_model.py:19-28; /root/reference/gaussian_renderer/__init__.py:216-227.
 * feat_hw [P,CF] channels-last compositor output (CF >= 9; channel 9 = accumulated depth when has_depth),
 * alphas [P], w1 [6,12], w2 [3,6]  ->  rgb [3,P] planar, depth [P].
 * Camera rays, one of:
 *   rays   [6,P] planar (the reference's cam_ray map, /root/reference/scene/cameras.py:132-146), ray_* = NULL;
 *   rays = NULL, ray_intr = device float[4] {fx, fy, cx, cy}, ray_c2w = device float[12] (the first three rows of
 *          the row-major camera-to-world matrix: a [3,4] or a [4,4] array), width = image width: the origin and
 *          the unit view direction through each pixel centre are generated in registers. */
int mobgs_decoder_fwd(int P, int CF, int has_depth, int width, const float* feat_hw, const float* alphas,
                      const float* rays, const float* ray_intr, const float* ray_c2w, const float* w1,
                      const float* w2, float* rgb, float* depth, void* stream);
/* w_partial: scratch [mobgs_decoder_bwd_blocks(P), 102] floats.  v_depth / v_rays may be NULL.
 * g_c2w (may be NULL): gradient of ray_c2w (in-kernel-ray mode only), g_c2w_floats = 12 ([3,4]) or 16 ([4,4]: the
 * fourth row is written as zeros).  accumulate_wgrad != 0: the weight gradients are added to g_w1 / g_w2 instead
 * of overwriting them (fixed summation order either way). */
int mobgs_decoder_bwd_blocks(int P);
int mobgs_decoder_bwd(int P, int CF, int has_depth, int width, const float* feat_hw, const float* alphas,
                      const float* rays, const float* ray_intr, const float* ray_c2w, const float* w1,
                      const float* w2, const float* v_rgb, const float* v_depth, float* v_feat_hw, float* v_alphas,
                      float* v_rays, float* w_partial, float* g_w1, float* g_w2, float* g_c2w, int g_c2w_floats,
                      int accumulate_wgrad, void* stream);

/* The decoder for a BATCH of C images in one launch (the K sub-frames of a blurry view): feat_hw [C,P,CF], alphas [C,P],
 * rgb [C,3,P], depth [C,P] (and the cotangents alike); *_stride = floats between consecutive images' ray maps /
 * intrinsics / poses, 0 = shared by all images.  The weight gradients are sums over all images (one fixed-order
 * reduction); g_c2w [C, g_c2w_floats], w_partial [C * mobgs_decoder_bwd_blocks(P), 102].  A shared ray map / pose
 * cannot receive a gradient when C > 1.  C = 1 is the single-image call. */
int mobgs_decoder_fwd_many(int C, int P, int CF, int has_depth, int width, const float* feat_hw, const float* alphas,
                           const float* rays, int64_t rays_stride, const float* ray_intr, int intr_stride,
                           const float* ray_c2w, int c2w_stride, const float* w1, const float* w2, float* rgb,
                           float* depth, void* stream);
int mobgs_decoder_bwd_many(int C, int P, int CF, int has_depth, int width, const float* feat_hw, const float* alphas,
                           const float* rays, int64_t rays_stride, const float* ray_intr, int intr_stride,
                           const float* ray_c2w, int c2w_stride, const float* w1, const float* w2, const float* v_rgb,
                           const float* v_depth, float* v_feat_hw, float* v_alphas, float* v_rays, float* w_partial,
                           float* g_w1, float* g_w2, float* g_c2w, int g_c2w_floats, int accumulate_wgrad,
                           void* stream);

/* The batched decoder with `n` further channels [c0, c0 + n) of the image -- behind the ones the decoder reads -- handed
 * out as an array of their own, chan_out [C,P,n] (forward), and their cotangent v_chan [C,P,n] written into those
 * channels of v_feat_hw (backward; the other unread channels get 0 as before).  get_flow()
 * (/root/reference/gaussian_renderer/__init__.py:436-476) composites 9 features + 2 flow channels in one pass; taking the
 * two channels off the 12-channel image with a PyTorch slice is a strided copy of the whole image each way.
 * chan_out / v_chan = NULL: exactly mobgs_decoder_fwd_many / mobgs_decoder_bwd_many. */
int mobgs_decoder_fwd_channels(int C, int P, int CF, int has_depth, int width, const float* feat_hw, const float* alphas,
                               const float* rays, int64_t rays_stride, const float* ray_intr, int intr_stride,
                               const float* ray_c2w, int c2w_stride, const float* w1, const float* w2, float* rgb,
                               float* depth, float* chan_out, int c0, int n, void* stream);
int mobgs_decoder_bwd_channels(int C, int P, int CF, int has_depth, int width, const float* feat_hw, const float* alphas,
                               const float* rays, int64_t rays_stride, const float* ray_intr, int intr_stride,
                               const float* ray_c2w, int c2w_stride, const float* w1, const float* w2, const float* v_rgb,
                               const float* v_depth, float* v_feat_hw, float* v_alphas, float* v_rays, float* w_partial,
                               float* g_w1, float* g_w2, float* g_c2w, int g_c2w_floats, int accumulate_wgrad,
                               const float* v_chan, int c0, int n, void* stream);

/* ---- K10: deformation network (the API the reference exposes as scene.deformation.deform_network) -----
 * /root/reference/scene/hexplane.py:75-108,165-187 (HexPlane multi-resolution bilinear planes, product over the
 * 6 planes of a level, concat over 3 levels -> 96 features); /root/reference/scene/deformation.py:158-199
 * (Linear 96->128, three ReLU-Linear(128,128)-ReLU-Linear(128,{7,3,4}) heads, point/scale/rotation update).
 *
 * planes_host / gplanes_host: HOST arrays of 18 device pointers, index level*6 + plane with plane order
 *   (0,1) (0,2) (0,3) (1,2) (1,3) (2,3); every plane is CHANNELS-LAST [rb][ra][32] (ra = resolution of the first
 *   axis of the pair, rb of the second) -- the host permutes the reference's [1,32,rb,ra] parameters.
 * ra_host / rb_host: HOST int32[18].   aabb: device [2,3] = {xyz_max, xyz_min} (the reference's convention).
 * mobgs_hexplane_fwd : pts [N,3], times [N]  ->  feat [N,96]
 * mobgs_hexplane_bwd : v_feat [N,96] -> gplanes (ACCUMULATED; zero them first), v_pts [N,3] (ADDED to its current
 *                      content), v_times [N] (written).  The (point, plane) contributions are counting-sorted by
 *                      plane cell (integer atomics) and runs of equal cells are summed in registers before they
 *                      leave as float atomics, instead of torch's one global float atomic per (point, plane, tap,
 *                      channel) (grid_sampler_2d_backward).  scratch: mobgs_hexplane_bwd_scratch_bytes(N, ra, rb)
 *                      bytes, 16-byte aligned (per-cell counters + entry lists of 144 B per point and plane).
 * mobgs_deform_mlp_fwd: feat + pts/scales/rots [N,3]/[N,3]/[N,4] -> out_pts, out_scales, out_rots (MFMA fp32).
 *   Weights K-major: W0t [96,128], b0 [128], W1t [3,128,128], b1 [3,128], W2t [3,128,32] (7/3/4 real columns,
 *   zero padded), b2 [3,32]; head order: position, scale, rotation.
 *   o_raw [N,16] (may be NULL): the 14 raw head outputs per point (position 0..6, scale 7..9, rotation 10..13),
 *   all the backward pass keeps of the forward pass.
 * mobgs_deform_mlp_bwd: replaces torch autograd over the reference's nn.Sequential heads
 *   (/root/reference/scene/deformation.py:56-73,158-199).  Recomputes the hidden activations from `feat`, then
 *   data gradients (v_feat [N,96], v_pts [N,3], v_rots [N,4], all WRITTEN; the scale input's cotangent is
 *   v_out_scales itself) and weight gradients, everything on fp32 MFMA.  Besides the K-major W0t / W1t it reads
 *   the weights in their original (out, in) layouts: W0 [128,96], W1 [3,128,128], W2pad [3,8,128] (rows >= 7/3/4
 *   zero).  v_out_* may be NULL (zero cotangent).  v_o: scratch [N,16] (cotangents of the raw head outputs).
 *   partials: scratch of mobgs_deform_mlp_bwd_blocks(N) x
 *   mobgs_deform_mlp_grad_floats() floats (one block of partial sums per workgroup, reduced in workgroup order:
 *   deterministic).  grads (mobgs_deform_mlp_grad_floats() = 66 080 floats, WRITTEN): W0 [128,96] | b0 [128] |
 *   W1 [3,128,128] | b1 [3,128] | W2 [32,128] (row 8h + o = output o of head h) | b2 [32] (same indexing). */
int mobgs_hexplane_fwd(int N, const float* pts, const float* times, const float* aabb,
                       const float* const* planes_host, const int32_t* ra_host, const int32_t* rb_host,
                       float* feat, void* stream);
int mobgs_hexplane_bwd(int N, const float* pts, const float* times, const float* aabb,
                       const float* const* planes_host, const int32_t* ra_host, const int32_t* rb_host,
                       const float* v_feat, float* const* gplanes_host, float* v_pts, float* v_times,
                       void* scratch, void* stream);
size_t mobgs_hexplane_bwd_scratch_bytes(int N, const int32_t* ra_host, const int32_t* rb_host);
int mobgs_deform_mlp_fwd(int N, const float* feat, const float* pts, const float* scales, const float* rots,
                         const float* W0t, const float* b0, const float* W1t, const float* b1, const float* W2t,
                         const float* b2, float* out_pts, float* out_scales, float* out_rots, float* o_raw,
                         void* stream);
int mobgs_deform_mlp_bwd_blocks(int N);
size_t mobgs_deform_mlp_grad_floats(void);
int mobgs_deform_mlp_bwd(int N, const float* feat, const float* pts, const float* rots, const float* o_raw,
                         const float* W0t, const float* b0, const float* W1t, const float* b1, const float* W0,
                         const float* W1, const float* W2pad, const float* v_out_pts, const float* v_out_scales,
                         const float* v_out_rots, float* v_feat, float* v_pts, float* v_rots, float* v_o,
                         float* partials, float* grads, void* stream);

/* ---- K10b: BLCE of one view (latent camera poses of the K = 9 sub-frames), forward and backward ----------
 * /root/reference/scene/blce.py:374-478 (BLCE.forward) + :150-152 (inverse of the warped poses), which the reference
 * runs as ~180 + ~350 tiny torch launches per view.  One single-wave kernel each way.
 * params_host / grads_host: HOST arrays of 22 device pointers in the order documented in csrc/blce.hip (the
 *   reference's parameter tensors of view `idx`: view_embedder table, Rt_encoder, view_encoder, blur_feature_encoder
 *   0/2/4, wv_derivative.{time_embedder, w_linear, v_linear}, rot/trans/theta decoders; weight then bias).
 * Rt [4,4] c2w of the view, blur_feature: device scalar.  c2w / w2c [9,4,4]: warped poses and their inverses.
 * saved: mobgs_blce_saved_floats() floats kept for the backward pass (NULL when no gradient is needed).
 * bwd: v_c2w / v_w2c [9,4,4] cotangents (either may be NULL); every gradient tensor is fully written (the
 *   view_embedder table gets zeros outside row idx). */
size_t mobgs_blce_saved_floats(void);
int mobgs_blce_fwd(const float* const* params_host, int idx, int num_views, const float* Rt,
                   const float* blur_feature, float* c2w, float* w2c, float* saved, void* stream);
int mobgs_blce_bwd(const float* const* params_host, float* const* grads_host, int idx, int num_views, const float* Rt,
                   const float* saved, const float* v_c2w, const float* v_w2c, void* stream);

/* ---- K11: fused photometric loss (L1 + SSIM), forward and backward -------------------------------------
 * /root/reference/utils/loss_utils.py:233-239 (l1_loss, mask=None), :251-260 + :351-381 (ssim: 11x11 Gaussian
 * window sigma 1.5, zero padding, per channel, mean over all elements); /root/reference/train.py:621-628.
 * img1, img2 [C,H,W].  fwd: partial [mobgs_ssim_l1_blocks(C,H,W), 2] = per-workgroup {sum ssim_map, sum |img1-img2|}
 * (the caller sums and divides by C*H*W); dmaps [3,C,H,W] = d ssim_map / d{mu1, E[x^2], E[xy]} (NULL when no
 * backward is needed).  bwd: scales = device float[C,2], per channel {d loss / d sum-of-ssim_map, d loss / d sum-of-|diff|}
 * (i.e. the mean's 1/(C*H*W) already applied); v_img1 [C,H,W] is fully written (gradient w.r.t. img1 only). */
int mobgs_ssim_l1_blocks(int C, int H, int W);
int mobgs_ssim_l1_fwd(int C, int H, int W, const float* img1, const float* img2, float* partial, float* dmaps,
                      void* stream);
int mobgs_ssim_l1_bwd(int C, int H, int W, const float* img1, const float* img2, const float* dmaps,
                      const float* scales, float* v_img1, void* stream);

/* ---- K17: flow-consistency loss (the consumer of get_flow()'s outputs in a training iteration) ---------------------
 * /root/reference/train.py:651-671 with utils/loss_utils.py:233-237 (masked l1_loss):
 *   loss = l1(grid_sample(ori expanded over K, norm(exp2mid)), latent, mask = latent_alpha)
 *        + l1(grid_sample(latent,             norm(mid2exp)), ori,    mask = d_alpha)
 * norm(c) = 2 c / (size - 1) - 1 per axis; grid_sample bilinear, padding_mode = 'border', align_corners = False;
 * l1(a, b, m) = sum |(a - b) m| / (3 sum(m) + 1e-8).  lambda_flow_loss is the caller's.
 * ori [B,3,H,W]; latent [B,K,3,H,W]; exp2mid, mid2exp [B,K,H,W,2] PIXEL coordinates (what get_flow returns; the
 * reference normalises them in place, this entry point leaves its inputs alone); latent_alpha [B,K,H,W]; d_alpha
 * [B,H,W].  H, W >= 2.
 * fwd: partial = scratch [mobgs_flow_warp_loss_blocks(B,H,W), 4]; sums [4] (kept for bwd) = {N1, S1, N2, S2};
 *      loss [1].  Deterministic (fixed summation order).
 * bwd: v_loss = device scalar.  g_ori [B,3,H,W] and g_latent [B,K,3,H,W] are ADDED to with float atomics (bilinear
 *      scatter, as torch's grid_sampler_2d_backward: summation order not fixed) -- zero-fill them; g_exp2mid /
 *      g_mid2exp [B,K,H,W,2], g_latent_alpha [B,K,H,W], g_d_alpha [B,H,W] are fully written.  Any may be NULL.
 *      scratch: mobgs_flow_warp_loss_bwd_scratch_floats(B,K,H,W) floats (needed when g_ori or g_latent is wanted: the
 *      images' gradients as TARGETS of the other term are written there and added at the end).
 *      combine_taps != 0: coinciding taps of neighbouring pixels (next lane, next row) are merged in registers before
 *      the atomics (same sums, a quarter of the atomics). */
int mobgs_flow_warp_loss_blocks(int B, int H, int W);
int mobgs_flow_warp_loss_fwd(int B, int K, int H, int W, const float* ori, const float* latent, const float* exp2mid,
                             const float* mid2exp, const float* latent_alpha, const float* d_alpha, float* partial,
                             float* sums, float* loss, void* stream);
int mobgs_flow_warp_loss_bwd(int B, int K, int H, int W, const float* ori, const float* latent, const float* exp2mid,
                             const float* mid2exp, const float* latent_alpha, const float* d_alpha, const float* sums,
                             const float* v_loss, float* g_ori, float* g_latent, float* g_exp2mid, float* g_mid2exp,
                             float* g_latent_alpha, float* g_d_alpha, float* scratch, int combine_taps, void* stream);
size_t mobgs_flow_warp_loss_bwd_scratch_floats(int B, int K, int H, int W);

/* ---- K16: one Adam step over MANY parameter tensors in one launch ------------------------------------------------
 * The reference steps three torch.optim.Adam optimisers per iteration (train.py:790-807: static Gaussians, dynamic
 * Gaussians + decoder, blur kernel), ~14 one-tensor parameter groups each: ~26 groups x 8 multi-tensor launches =
 * 3.3 ms of device time per iteration at 300 k Gaussians for 0.48 GB of traffic.  One launch over all of them:
 *   m = m + (1 - beta1) (g - m);  v = beta2 v + (1 - beta2) g^2;  p = p - step_size * m / (sqrt(v) / bias2_sqrt + eps)
 * (torch.optim.Adam, amsgrad = False, weight_decay = 0, maximize = False; step_size = lr / (1 - beta1^t) and
 * bias2_sqrt = sqrt(1 - beta2^t) are computed by the caller per tensor, as torch does on the host; the betas arrive as
 * doubles and 1 - beta is formed in double before it is rounded to fp32, again as torch does: 1.f - 0.999f is off by 1e-5).
 * tensors_host: HOST array of n_tensors descriptors (device pointers, fp32, contiguous); n_tensors <= 64. */
typedef struct MobgsAdamTensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    float step_size;
    float bias2_sqrt;
} MobgsAdamTensor;
int mobgs_adam_step(int n_tensors, const MobgsAdamTensor* tensors_host, double beta1, double beta2, double eps,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOBGS_HIP_H */
