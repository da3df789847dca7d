// K3-K5: tile intersection lists for gfx950 -- count, offsets, emit, per-tile depth sort.
//
// Replaces gsplat v1.4.0 isect_tiles (two passes) + CUB DeviceRadixSort::SortPairs + isect_offset_encode
// [upstream, SURVEY.md Appendix A.2], which run inside every rasterization() call of the reference
// (/root/reference/gaussian_renderer/__init__.py:143, :163, :201, :236, :255, :274, ...).
//
// MI355X design: instead of one global 64-bit radix sort over all I intersections (6-8 passes x 24 B x I
// of HBM traffic), intersections are binned straight into tile-contiguous segments (one histogram pass,
// one scatter pass, both load-balanced one-thread-per-intersection) and each tile's segment is sorted
// inside LDS by a bitonic network on the 64-bit key (depth bits << 32 | flat splat id).  The key is
// unique, so the unstable network reproduces exactly the order of upstream's stable radix sort:
// ascending depth bits, ties by ascending splat index.
//
// Reach culling (optional, on by default in the host wrapper): upstream lists a splat in every tile its
// 3-sigma bounding BOX touches.  A (tile, splat) pair whose smallest possible sigma over the tile's pixel
// rectangle already exceeds ln(255 * opacity) cannot reach alpha >= 1/255 at any pixel of the tile, so the
// compositor would skip it at all 256 pixels; dropping the pair from the list leaves every pixel bit-identical (gradients: same terms)
// and removes ~half of the intersections on anisotropic scenes.  The test is conservative (margin on the
// threshold); with culling off the lists are exactly upstream's.
#include "common.h"

namespace mobgs {

// ---------------------------------------------------------------------------------------------------
// single-pass exclusive scans (chained scan with decoupled look-back)
//
// Each workgroup owns one chunk of SCAN_BLOCK consecutive elements.  Chunks are handed out through an atomic
// ticket, so a workgroup only ever waits for chunks whose workgroups are already running (forward progress
// does not depend on the dispatch order).  A workgroup publishes its chunk total in a 64-bit status word
// {flag, value}, walks back over its predecessors until it meets one that already knows its inclusive prefix,
// and publishes its own.  The status words and tickets live in the caller's scratch and are zeroed by the one
// memset that also clears the per-tile counters.  The status word IS the message (no other memory is handed from
// workgroup to workgroup), so relaxed agent-scope atomics suffice -- release/acquire at agent scope would write
// back / invalidate the whole XCD L2 on every access (measured: 10x slower kernels).
// ---------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 512;  // x 4 items: shorter per-thread chains than 256 x 8 (-6 us in bin), same 2048-element chunks
                                   // (1024 x 4 = 4096-element chunks, fewer counter atomics per entry: bin 49 -> 61 us)
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;  // 2048 elements per workgroup

__device__ inline int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(v, off, 64);
        if (lane >= off) v += n;
    }
    return v;
}

// inclusive scan across a workgroup of NW waves; returns this thread's inclusive value, *total = block sum
template <int NW>
__device__ inline int block_incl_scan_w(int v, int* total) {
    __shared__ int wsum[NW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int inc = wave_incl_scan(v, lane);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        base += (k < wv) ? wsum[k] : 0;
        tot += wsum[k];
    }
    *total = tot;
    __syncthreads();
    return inc + base;
}
__device__ inline int block_incl_scan(int v, int* total) { return block_incl_scan_w<SCAN_THREADS / 64>(v, total); }

constexpr uint64_t LB_AGGREGATE = 1ull << 32, LB_PREFIX = 2ull << 32;

__device__ inline int take_ticket(int32_t* counter) {
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(counter, 1);
    __syncthreads();
    return s_ticket;
}

// exclusive prefix of chunk `chunk` given its total `aggregate` (uniform over the workgroup).  The first wave
// inspects 64 predecessors per step: it needs every status word down to the nearest one that already holds an
// inclusive prefix, sums those and stops there (a serial walk costs one L2 round trip per predecessor).
__device__ inline int lookback_exclusive(uint64_t* status, int chunk, int aggregate) {
    __shared__ int s_excl;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int excl = 0;
        if (chunk > 0) {
            if (lane == 0)
                __hip_atomic_store(&status[chunk], LB_AGGREGATE | (uint32_t)aggregate, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            int top = chunk - 1;  // lane l looks at chunk top - l; chunk "-1" counts as prefix 0
            for (;;) {
                const int idx = top - lane;
                const uint64_t w = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                            : LB_PREFIX;
                const uint64_t not_ready = __builtin_amdgcn_ballot_w64((w >> 32) == 0);
                const uint64_t is_prefix = __builtin_amdgcn_ballot_w64((w >> 32) == 2);
                const int first = is_prefix ? __builtin_ctzll(is_prefix) : 64;
                const uint64_t needed = first >= 63 ? ~0ull : ((2ull << first) - 1);
                if (not_ready & needed) continue;  // a needed predecessor has not published yet: look again
                int v = lane <= first ? (int)(uint32_t)w : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                excl += v;
                if (first < 64) break;
                top -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&status[chunk], LB_PREFIX | (uint32_t)(excl + aggregate), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
        }
    }
    __syncthreads();
    return s_excl;
}

// cum[i] = sum of in[0..i), cum[n] = total (also stats_slot); n > 0.
// chunk_owner[m] (m < owner_slots) = the element whose interval [cum[i], cum[i+1]) holds position m * KEEP_CHUNK, for
// every such position below the total: bin_kernel's workgroup m starts from it instead of searching cum (a chain of
// dependent loads of data another XCD has just written, at the head of every workgroup).
__global__ void __launch_bounds__(SCAN_THREADS) scan_lookback_kernel(int n, const int32_t* __restrict__ in,
                                                                       int32_t* __restrict__ cum,
                                                                       int32_t* __restrict__ counter,
                                                                       uint64_t* __restrict__ status,
                                                                       int64_t* __restrict__ stats_slot,
                                                                       int32_t* __restrict__ chunk_owner,
                                                                       int owner_slots,
                                                                       const int32_t* __restrict__ order,
                                                                       int32_t* __restrict__ cum_enum) {
    // order != NULL (round 5, fused path): the bounding-box intersections are ENUMERATED splat order[0], order[1], ...
    // instead of 0, 1, ... -- a caller-chosen permutation of the splats (spatially coherent: neighbouring chunks of
    // intersections then hit neighbouring tiles, and bin_kernel's LDS ranking turns ~1000 returning atomics per
    // workgroup into ~100).  cum_enum[k] = first intersection of the k-th splat of that order (what bin_kernel searches);
    // cum[g] = first intersection of splat g (what the backward pass and the slot reduction look up).  Every consumer
    // only relies on "the intersections of a splat are consecutive": list contents, list order and the per-splat
    // order of the gradient slots do not depend on the permutation.
    const int chunk = take_ticket(counter);
    const int base = chunk * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS], gid[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        gid[k] = (order && base + k < n) ? order[base + k] : base + k;
        v[k] = (base + k < n) ? in[gid[k]] : 0;
        s += v[k];
    }
    int total;
    const int inc = block_incl_scan(s, &total);
    int run = lookback_exclusive(status, chunk, total) + inc - s;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) {
            cum[gid[k]] = run;
            if (order) cum_enum[base + k] = run;
        }
        if (v[k] > 0) {  // chunk starts inside [run, run + v[k]): none or one, more only for splats over > 2048 tiles
            const unsigned m1 = ((unsigned)run + (unsigned)v[k] - 1u) >> KEEP_CHUNK_LOG2;
            for (unsigned m = ((unsigned)run + (unsigned)KEEP_CHUNK - 1u) >> KEEP_CHUNK_LOG2;
                 m <= m1 && m < (unsigned)owner_slots; ++m)
                chunk_owner[m] = base + k;
        }
        run += v[k];
    }
    if (base <= n - 1 && n - 1 < base + SCAN_ITEMS) {  // the thread that owns the last element
        cum[n] = run;
        if (order) cum_enum[n] = run;
        if (stats_slot) *stats_slot = (int64_t)run;
    }
}

// ---------------------------------------------------------------------------------------------------
// pass A: one workgroup per chunk of SCAN_BLOCK consecutive bounding-box intersections
// ---------------------------------------------------------------------------------------------------
// largest g in [lo, hi) with cum[g] <= j   (cum is non-decreasing, cum[lo] <= j < cum[hi])
template <typename Cum>
__device__ inline int owner_in(const Cum& cum, int lo, int hi, int j) {
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] <= j)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// The same answer found by a whole wave: every lane probes one of 64 evenly spaced candidates per round, the ballot of
// "cum[probe] <= j" (a prefix of ones, cum is non-decreasing) narrows [lo, hi) 64-fold -- 4 dependent global loads for
// 300 k splats where the binary search above needs 18.  bin_kernel's two chunk-boundary searches sit at the head of
// every workgroup's dependency chain (2 x 18 L2 round trips before the first useful load).  All lanes must be
// active and call it with the same arguments.
__device__ inline int owner_in_wave(const int32_t* __restrict__ cum, int lo, int hi, int j) {
    const int lane = threadIdx.x & 63;
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) >> 6;
        const int probe = lo + (lane + 1) * step;
        const bool le = probe < hi && cum[probe] <= j;
        const int c = __builtin_popcountll(__builtin_amdgcn_ballot_w64(le));
        lo += c * step;
        hi = min(hi, lo + step);
    }
    return lo;
}

// stride of the per-tile list counters (1 = packed; giving each counter its own 128-byte line was measured: no gain
// for the atomics of a dense image region, +12 us in tile_scan)
constexpr int TC_STRIDE = 1;
// The rank counters exist TC_COPIES times ([copy][tile]); workgroup (chunk) c of bin_kernel uses copy c mod TC_COPIES.
// Device-scope atomics on one address are served one after the other at the memory side of the chip (~120 ns each:
// the L2s of the eight XCDs are not coherent with each other); every tile counter receives one atomic from almost
// every chunk that touches the tile -- ~200 per counter at 300 k splats, 150 at 30 k -- and that queue was 42 % /
// 62 % of bin_kernel (ablated build: 59.1 -> 34.4 us, 29.8 -> 11.3 us).  With the copies a counter's queue is 8 x
// shorter; tile_scan_kernel sums the copies into the list lengths and leaves every (copy, tile) pair's first
// position in tile_base, which is what emit_kernel adds the rank to.  Ranks only have to be distinct inside a list.
constexpr int TC_COPIES = 8;
// (Tried in round 5 and dropped: one counter copy per XCD -- the copy chosen by HW_REG_XCC_ID, each in its own cache
// lines -- with workgroup-scope atomics, hoping to have them performed in the XCD's own L2.  gfx950 emits the same
// global_atomic_add for either scope and the kernel's time did not move: bin_kernel is bound by the RATE of returning
// atomics + scattered key stores, ~3.4 M transactions per launch, not by their scope.)
__device__ __forceinline__ int rank_add(int32_t* p, int v) { return atomicAdd(p, v); }
constexpr int OWNER_LDS = 4096;  // cum_tiles entries of the chunk's owner range cached in LDS
static_assert(SCAN_BLOCK == KEEP_CHUNK, "one workgroup per keep_scan chunk");

// Owner splat (search in the chunk's slice of cum_tiles, cached in LDS), tile and reach test of every
// intersection of the chunk.  A kept intersection takes its rank inside the tile's list with ONE returning
// atomic.  The keep flags are scanned inside the chunk (keep_scan locals, see common.h); the chunk's total goes to its
// base word, which tile_scan_kernel turns into the exclusive prefix over the chunks.  (owner, tile, rank) of the KEPT
// intersections are written out compacted inside the chunk's own range [chunk * 2048, + kept) with the count in
// chunk_cnt, so that pass B is a pure streaming scatter that reads 12 bytes per listed entry instead of 16 per
// bounding-box intersection.
//
// DENSE variant (grids of up to DENSE_MAX_TILES tiles): a dense image region sends thousands of rank atomics to the
// same few counters, which serialise in L2 (bin 73 -> 249 us on scripts/heavy_tail.py).  The workgroup first ranks
// its kept intersections per tile in LDS (one int per tile, dynamic shared memory), then ONE thread per touched tile
// reserves the workgroup's range with a single global atomic: up to 20x fewer same-address atomics there, and still
// ahead on uniform scenes (fewer returning global atomics per chunk).
//
// FUSED variant (round 5; single-pass lists): the splat's position, conic, reach threshold, box origin / width and
// depth bits come from ONE 48-byte bin record the projection kernel left behind (common.h, write_bin_record) instead
// of five gathers, two divisions and a logarithm per intersection, and the kept intersection's 64-bit sort key goes
// STRAIGHT into the tile's segment of a strided key arena -- [tile][counter copy][seg_stride] keys -- at the rank the
// atomic returned: no (owner, tile, rank) triples, no per-(copy, tile) base table, no emit pass.  A rank beyond
// seg_stride is dropped; tile_finish_kernel sees the counter and hands every consumer empty lists (the caller then
// falls back to the two-pass path with the true counts).
template <bool DENSE, bool FUSED>
__global__ void __launch_bounds__(SCAN_THREADS)
bin_kernel(int n_gauss, int N, int tile_w, int tile_h, int width, int height, int cull, int capacity,
           const int32_t* __restrict__ cum, const float* __restrict__ means2d, const int32_t* __restrict__ radii,
           const float* __restrict__ conics, const float* __restrict__ opacities, int opac_per_camera,
           int32_t* __restrict__ chunk_cnt, int32_t* __restrict__ owner, int32_t* __restrict__ tile_of_j,
           int32_t* __restrict__ rank_of_j, int32_t* __restrict__ tile_count, int32_t* __restrict__ keep_scan,
           int n_tiles_total, const int32_t* __restrict__ chunk_owner, const float* __restrict__ binrec,
           uint64_t* __restrict__ seg_keys, int seg_stride, int count_stride, const int32_t* __restrict__ order,
           int dense_window) {
    __shared__ int s_cum[OWNER_LDS + 1];
    // DENSE: per-tile count of this workgroup, then the base of its range -- for the `dense_window` tiles of ONE camera
    // (round 6: a batch of C cameras has C x the tiles, but the 2048 intersections of a chunk belong to one camera -- the
    // enumeration is camera-major -- so the table covers the camera of the chunk's first intersection; the few
    // intersections of a chunk that straddles two cameras take the direct atomic.  8 cameras at 1352x1014: 43 520 tiles
    // did not fit the 8192-entry table, the batch fell back to direct atomics: 357 us against 8 x 30)
    extern __shared__ int s_tile[];
    const int chunk = blockIdx.x;
    int32_t* kchunk = keep_scan + (size_t)chunk * (KEEP_CHUNK + 1);
    // this workgroup's copy of the counters (fused path: copies padded to whole 128-byte lines, count_stride apart)
    const int copy = chunk & (TC_COPIES - 1);
    tile_count += (size_t)copy * (FUSED ? count_stride : n_tiles_total);
    const int I = min(cum[n_gauss], capacity);
    const int start = chunk * SCAN_BLOCK;
    if (start >= I) {
        if (threadIdx.x == 0) {
            kchunk[0] = 0;                   // empty chunk
            if (!FUSED) chunk_cnt[chunk] = 0;
            if (start == I) kchunk[1] = 0;  // local of position I (one past the last intersection)
        }
        return;
    }
    const int end = min(I, start + SCAN_BLOCK);
    // owner of the chunk's first intersection: left by the scan.  Upper end of the owner range: the owner of the NEXT
    // chunk's first intersection when there is one (>= the owner of this chunk's last, which is all the searches
    // below need: cum[g_hi + 1] > every j of the chunk), else -- one workgroup per launch -- a wave-wide search
    const int g_lo = chunk_owner[chunk];
    const int g_hi = end < I ? chunk_owner[chunk + 1] : owner_in_wave(cum, g_lo, n_gauss, end - 1);
    const int span = g_hi - g_lo + 1;
    const bool cached = span <= OWNER_LDS;
    if (cached) {
        for (int t = threadIdx.x; t <= span; t += SCAN_THREADS) s_cum[t] = cum[g_lo + t];
        __syncthreads();
    }
    const int tiles_per_cam = tile_w * tile_h;
    // item k of thread t is intersection start + k * SCAN_THREADS + t: neighbouring lanes work on neighbouring
    // intersections (coalesced stores, shared owner data)
    __shared__ int s_cnt[SCAN_ITEMS][SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int keep[SCAN_ITEMS], own[SCAN_ITEMS], til[SCAN_ITEMS], before[SCAN_ITEMS];
    uint32_t dbits[SCAN_ITEMS];  // FUSED: depth bits of the item's splat (high half of its sort key)
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int j = start + k * SCAN_THREADS + threadIdx.x;
        int kp = 0, g = 0, t = 0;
        dbits[k] = 0u;
        if (FUSED && j < end) {
            // (`cum` is the scan in ENUMERATION order here: position ge of the caller's splat order, or the splat itself)
            const int ge = cached ? g_lo + owner_in(s_cum, 0, span, j) : owner_in(cum, g_lo, g_hi + 1, j);
            g = order ? order[ge] : ge;
            const float4* rec = reinterpret_cast<const float4*>(binrec + (size_t)g * BIN_RECORD_FLOATS);
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
            const int q = j - (cached ? s_cum[ge - g_lo] : cum[ge]);
            const unsigned wc = __float_as_uint(r2.z);
            const int w = (int)(wc & 0xFFFFu);
            const unsigned xy0 = __float_as_uint(r1.w);
            // q / w as below (exact for q < 2^21)
            const int row = (int)(((float)q + 0.5f) * __builtin_amdgcn_rcpf((float)w));
            const int ty = (int)(xy0 >> 16) + row, tx = (int)(xy0 & 0xFFFFu) + (q - row * w);
            t = (int)(wc >> 16) * tile_w * tile_h + ty * tile_w + tx;
            const float x0 = (float)(tx * MOBGS_TILE) + 0.5f, y0 = (float)(ty * MOBGS_TILE) + 0.5f;
            const float x1 = fminf((float)(tx * MOBGS_TILE) + 15.5f, (float)width - 0.5f);
            const float y1 = fminf((float)(ty * MOBGS_TILE) + 15.5f, (float)height - 0.5f);
            // threshold -1: never listed (min sigma >= 0); REACH_ALWAYS: listed whatever the arithmetic below yields
            kp = (r0.z >= REACH_ALWAYS ||
                  min_sigma_over_tile_pre(r0.x, r0.y, r1.x, r1.y, r1.z, r2.x, r2.y, x0, x1, y0, y1) <= r0.z) ? 1 : 0;
            dbits[k] = __float_as_uint(r0.w);
        }
        if (!FUSED && j < end) {
            g = cached ? g_lo + owner_in(s_cum, 0, span, j) : owner_in(cum, g_lo, g_hi + 1, j);
            const float2 m = reinterpret_cast<const float2*>(means2d)[g];
            const TileRect tr = tile_rect(m.x, m.y, radii[g], tile_w, tile_h);
            const int q = j - (cached ? s_cum[g - g_lo] : cum[g]);
            const int w = tr.x1 - tr.x0;
            // q / w for 0 <= q < w * h (at most the camera's tile grid), w >= 1: (q + 0.5) / w is at least 0.5 / w away
            // from an integer and the 1-ulp reciprocal + product move it by < 2.4e-7 * q / w, so the truncation is
            // exact for q < 2^21 (a 23 k x 23 k image) -- at a quarter of the instructions of the integer division
            const int row = (int)(((float)q + 0.5f) * __builtin_amdgcn_rcpf((float)w));
            const int ty = tr.y0 + row, tx = tr.x0 + (q - row * w);
            t = (g / N) * tiles_per_cam + ty * tile_w + tx;
            kp = 1;
            if (cull) {
                const float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
                const float op = opacities[opac_per_camera ? g : g % N];
                if (!(op * 255.f >= 1.f)) {
                    kp = 0;  // alpha = min(0.999, op * exp(-sigma)) < 1/255 everywhere (sigma >= 0 where blended)
                } else if (ca > 0.f && cc > 0.f) {
                    const float x0 = (float)(tx * MOBGS_TILE) + 0.5f, y0 = (float)(ty * MOBGS_TILE) + 0.5f;
                    const float x1 = fminf((float)(tx * MOBGS_TILE) + 15.5f, (float)width - 0.5f);
                    const float y1 = fminf((float)(ty * MOBGS_TILE) + 15.5f, (float)height - 0.5f);
                    kp = (min_sigma_over_tile(m.x, m.y, ca, cb, cc, x0, x1, y0, y1) <= reach_threshold(op)) ? 1 : 0;
                }
            }
        }
        keep[k] = kp;
        own[k] = g;
        til[k] = t;
        const uint64_t ballot = __builtin_amdgcn_ballot_w64(kp != 0);
        before[k] = __builtin_popcountll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) s_cnt[k][wv] = __builtin_popcountll(ballot);
    }
    // ranks inside the tiles' lists: all returning atomics in flight before the first result is consumed
    int rank[SCAN_ITEMS];
    if (DENSE) {
        __shared__ int s_wbase;
        if (threadIdx.x == 0) s_wbase = (til[0] / dense_window) * dense_window;  // (item 0 of thread 0 = intersection `start`)
        for (int t = threadIdx.x; t < dense_window; t += SCAN_THREADS) s_tile[t] = 0;
        __syncthreads();
        const int wbase = s_wbase;
        int loc[SCAN_ITEMS];   // index into the table, or -1: outside the window (direct atomic)
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            const int l = til[k] - wbase;
            loc[k] = (keep[k] && (unsigned)l < (unsigned)dense_window) ? l : -1;
            rank[k] = loc[k] >= 0 ? atomicAdd(&s_tile[loc[k]], 1)
                                  : (keep[k] ? rank_add(&tile_count[til[k] * TC_STRIDE], 1) : 0);
        }
        __syncthreads();
        int base_of[SCAN_ITEMS];
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k)  // the intersection that got local rank 0 speaks for its tile
            base_of[k] = (loc[k] >= 0 && rank[k] == 0) ? rank_add(&tile_count[til[k] * TC_STRIDE], s_tile[loc[k]]) : 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k)
            if (loc[k] >= 0 && rank[k] == 0) s_tile[loc[k]] = base_of[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k)
            if (loc[k] >= 0) rank[k] += s_tile[loc[k]];
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) rank[k] = keep[k] ? rank_add(&tile_count[til[k] * TC_STRIDE], 1) : 0;
    }
    __syncthreads();
    // exclusive prefix of every (item row, wave) segment in intersection order, and the chunk total
    int seg[SCAN_ITEMS];
    int total = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
#pragma unroll
        for (int w2 = 0; w2 < SCAN_THREADS / 64; ++w2) {
            if (w2 == wv) seg[k] = total;
            total += s_cnt[k][w2];
        }
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int i = k * SCAN_THREADS + threadIdx.x;
        if (start + i < end) {
            const int local = seg[k] + before[k];
            kchunk[1 + i] = local;
            if (FUSED) {
                if (keep[k] && rank[k] < seg_stride)
                    seg_keys[((size_t)til[k] * TC_COPIES + copy) * (size_t)seg_stride + rank[k]] =
                        ((uint64_t)dbits[k] << 32) | (uint32_t)own[k];
            } else if (keep[k]) {  // compacted: the chunk's kept intersections in order, at the start of its own range
                owner[start + local] = own[k];
                tile_of_j[start + local] = til[k];
                rank_of_j[start + local] = rank[k];
            }
        }
    }
    if (threadIdx.x == 0) {
        if (!FUSED) chunk_cnt[chunk] = total;
        kchunk[0] = total;
        if (end == I && end - start < SCAN_BLOCK) kchunk[1 + (end - start)] = total;  // local of position I
    }
}

// single workgroup: exclusive scan of tile_count[nt] -> tile_offsets[nt+1]; stats[2] = max count;
// tile_order[sched_slots(nt)] (optional) = the tiles by DESCENDING list length (counting sort on ORDER_BUCKETS length
// classes): the order in which the compositing kernels hand tiles to workgroups, so that the longest lists start
// first and the waves of one workgroup get lists of similar length (longest-processing-time-first scheduling).
// Tiles whose list is at least `heavy_len` long (at most nt/8 of them, the longest) take a whole workgroup: their
// id is written with SCHED_HEAVY into 4 consecutive slots and each of the 4 waves composites one 8x8 quadrant, so
// that one very long list does not become the critical path of the launch.  Only a schedule: images are
// bit-identical for any permutation / heavy marking, gradients equal up to the summation order of the quadrants.
constexpr int ORDER_BUCKETS = 1024;
constexpr int ORDER_LDS_TILES = 8192;  // list lengths the order workgroup keeps in LDS (32 KiB); beyond: re-read
constexpr int TSCAN_THREADS = 1024;
__device__ __forceinline__ int tile_total(const int32_t* __restrict__ tile_count, int nt, int i) {
    int v = 0;
#pragma unroll
    for (int c = 0; c < TC_COPIES; ++c) v += tile_count[(size_t)c * nt + i];
    return v;
}

__global__ void __launch_bounds__(TSCAN_THREADS) tile_scan_kernel(int nt, const int32_t* __restrict__ tile_count,
                                                                    int32_t* __restrict__ tile_base,
                                                                    int32_t* __restrict__ tile_offsets,
                                                                    int64_t* __restrict__ stats,
                                                                    int32_t* __restrict__ tile_order,
                                                                    int64_t capacity_box, int64_t capacity_listed,
                                                                    int32_t* __restrict__ keep_scan, int n_chunks,
                                                                    int heavy_len, int64_t* stats_mirror,
                                                                    int64_t stats_seq) {
    __shared__ int smax[TSCAN_THREADS / 64];
    __shared__ int hist[ORDER_BUCKETS];
    __shared__ int s_len[ORDER_LDS_TILES];  // list lengths (sum over the counter copies) of the order workgroup
    auto length_of = [&](int i) { return i < ORDER_LDS_TILES ? s_len[i] : tile_total(tile_count, nt, i); };
    // workgroup 1: chunk totals -> chunk bases of keep_scan (bin_kernel left each chunk's total in its base
    // word); stats[1] = I_listed.  Workgroup 0 does the per-tile work below at the same time.
    if (blockIdx.x == 1) {
        int base = 0;
        for (int c0 = 0; c0 < n_chunks; c0 += TSCAN_THREADS) {
            const int c = c0 + threadIdx.x;
            int32_t* w = keep_scan + (size_t)c * (KEEP_CHUNK + 1);
            const int v = (c < n_chunks) ? *w : 0;
            int total;
            const int inc = block_incl_scan_w<TSCAN_THREADS / 64>(v, &total);
            if (c < n_chunks) *w = base + inc - v;
            base += total;
        }
        if (threadIdx.x == 0) stats[1] = (int64_t)base;
        return;
    }
    // workgroup 2 (launched when a schedule is wanted): the tile order, concurrently with the offsets of workgroup 0
    const bool order_wg = blockIdx.x == 2;
    int carry = 0, mx = 0;
    for (int start = 0; start < nt; start += TSCAN_THREADS) {
        const int i = start + threadIdx.x;
        int cnt[TC_COPIES];
        int v = 0;
#pragma unroll
        for (int c = 0; c < TC_COPIES; ++c) {
            cnt[c] = (i < nt) ? tile_count[(size_t)c * nt + i] : 0;
            v += cnt[c];
        }
        mx = max(mx, v);
        if (order_wg) {  // only needs the longest list here; keeps the lengths for its two passes below
            if (i < ORDER_LDS_TILES) s_len[i] = v;
            continue;
        }
        int total;
        const int inc = block_incl_scan_w<TSCAN_THREADS / 64>(v, &total);
        if (i < nt) {
            int at = carry + inc - v;
            tile_offsets[i] = at;
#pragma unroll
            for (int c = 0; c < TC_COPIES; ++c) {  // first position of the entries ranked through copy c
                tile_base[(size_t)c * nt + i] = at;
                at += cnt[c];
            }
        }
        carry += total;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mx;
    __syncthreads();
    int longest = 0;
#pragma unroll
    for (int k = 0; k < TSCAN_THREADS / 64; ++k) longest = max(longest, smax[k]);
    if (!order_wg && threadIdx.x == 0) {
        tile_offsets[nt] = carry;
        stats[2] = (int64_t)longest;
        // the host's copy of {I_box, I_listed, longest list}: written straight into its pinned, device-mapped slot
        // (visible to the host once the event recorded behind this kernel has completed) -- no copy kernel
        if (stats_mirror) {
            stats_mirror[0] = stats[0];
            stats_mirror[1] = (int64_t)carry;
            stats_mirror[2] = (int64_t)longest;
            __threadfence_system();
            // sequence number LAST: a host that polls this word needs no event (no marker packet in the queue)
            if (stats_seq) {
                __hip_atomic_store(&stats_mirror[3], stats_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    // Arena too small (only checked when the caller runs ahead of the read-back, capacity_listed > 0): hand every
    // consumer EMPTY lists, so that kernels already enqueued behind this one touch nothing; stats keep the true
    // counts and the host redoes the binning with a larger arena.
    if (!order_wg) {
        if (capacity_listed > 0 && (stats[0] > capacity_box || (int64_t)carry > capacity_listed)) {
            __syncthreads();
            for (int i = threadIdx.x; i <= nt; i += TSCAN_THREADS) tile_offsets[i] = 0;
        }
        return;
    }
    if (!tile_order) return;
    auto bucket = [&](int len) {
        const int q = longest > 0 ? (int)(((int64_t)len * (ORDER_BUCKETS - 1)) / longest) : 0;
        return ORDER_BUCKETS - 1 - q;  // bucket 0 = the longest lists
    };
    static_assert(ORDER_BUCKETS == TSCAN_THREADS, "one histogram bin per thread");
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += TSCAN_THREADS) atomicAdd(&hist[bucket(length_of(i))], 1);
    __syncthreads();
    {
        const int mine = hist[threadIdx.x];
        int total;
        const int inc = block_incl_scan_w<TSCAN_THREADS / 64>(mine, &total);
        hist[threadIdx.x] = inc - mine;
    }
    __syncthreads();
    // heavy = every tile in a length class above class B, where B is the class `heavy_len` falls into, lowered until
    // at most sched_max_heavy tiles qualify: the heavy SET depends only on the tiles' classes (deterministic -- the
    // atomic order inside a class must not decide who is heavy, or gradients would differ from run to run)
    __shared__ int s_heavy, s_cut;
    if (threadIdx.x == 0) s_cut = 0;
    __syncthreads();
    if (heavy_len > 0 && longest >= heavy_len) {
        const int b_thr = bucket(heavy_len);
        if ((int)threadIdx.x <= b_thr && hist[threadIdx.x] <= (int)sched_max_heavy((size_t)nt))
            atomicMax(&s_cut, (int)threadIdx.x);
    }
    __syncthreads();
    if (threadIdx.x == 0) s_heavy = hist[s_cut];  // tiles in the classes before the cut (hist = exclusive prefix)
    const int n_slots = (int)sched_slots((size_t)nt);
    for (int i = threadIdx.x; i < n_slots; i += TSCAN_THREADS) tile_order[i] = -1;
    __syncthreads();
    const int n_heavy = s_heavy;
    for (int i = threadIdx.x; i < nt; i += TSCAN_THREADS) {
        const int pos = atomicAdd(&hist[bucket(length_of(i))], 1);
        if (pos < n_heavy) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tile_order[4 * pos + q] = i | SCHED_HEAVY;
        } else {
            tile_order[3 * n_heavy + pos] = i;
        }
    }
}

// Pass B: streaming scatter of the 64-bit sort keys into the tile-contiguous segments
__global__ void __launch_bounds__(1024) emit_kernel(const int32_t* __restrict__ n_box_ptr,
                                                     const int32_t* __restrict__ chunk_cnt,
                                                     const int32_t* __restrict__ owner,
                                                     const int32_t* __restrict__ tile_of_j,
                                                     const int32_t* __restrict__ rank_of_j,
                                                     const float* __restrict__ depths,
                                                     const int32_t* __restrict__ tile_base, int nt,
                                                     uint64_t* __restrict__ sort_keys, int capacity,
                                                     const int64_t* __restrict__ stats, int64_t capacity_listed) {
    // speculative launch (stats != NULL): nothing to do when the arena was too small (see tile_scan_kernel)
    if (stats && (stats[0] > (int64_t)capacity || stats[1] > capacity_listed)) return;
    const int I = min(*n_box_ptr, capacity);
    const int n_chunks = (I + KEEP_CHUNK - 1) >> KEEP_CHUNK_LOG2;
    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int cnt = chunk_cnt[chunk];
        for (int l = threadIdx.x; l < cnt; l += 1024) {  // two rounds per chunk at most
            const int j = (chunk << KEEP_CHUNK_LOG2) + l;
            const int g = owner[j];
            const uint32_t db = __float_as_uint(depths[g]);
            const int32_t* base = tile_base + (size_t)(chunk & (TC_COPIES - 1)) * nt;  // the copy bin_kernel ranked in
            sort_keys[(size_t)base[tile_of_j[j]] + rank_of_j[j]] = ((uint64_t)db << 32) | (uint32_t)g;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// per-tile bitonic sort ("mirror" formulation: every compare-exchange is ascending, so elements past the
// end behave as +inf without being stored and any n works)
//
// Barrier economy: every step of a merge of size k only touches its own aligned k-block, and the steps with
// compare distance j <= 64 only touch aligned 128-element chunks.  Each wave owns whole 128-element chunks
// (64 compare-exchanges per step = one per lane), so all those steps need no workgroup barrier -- the LDS
// queue of a wave is in order.  Only the mirror step and the j >= 128 steps of merges k >= 256 synchronise the
// workgroup: 6 barriers instead of 45 for a 512-entry list.
// ---------------------------------------------------------------------------------------------------
constexpr int CHUNK = 128;

__device__ __forceinline__ void cmpx(uint64_t* a, int lo, int hi, int n) {
    if (hi < n) {
        const uint64_t x = a[lo], y = a[hi];
        if (x > y) {
            a[lo] = y;
            a[hi] = x;
        }
    }
}

__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// steps j = j_first .. 1 (halving) inside every 128-chunk owned by this wave; no workgroup barrier
__device__ __forceinline__ void local_halving_steps(uint64_t* a, int n, int n2, int j_first, int wave, int nwaves,
                                                    int lane) {
    for (int c = wave * CHUNK; c < n2 && c < n; c += nwaves * CHUNK) {
        for (int j = j_first; j >= 1; j >>= 1) {
            const int blk = lane / j, l = lane - blk * j;
            const int lo = c + blk * 2 * j + l;
            cmpx(a, lo, lo + j, n);
            wave_lds_order();
        }
    }
}

template <int THREADS>
__device__ inline void bitonic_sort_lds(uint64_t* a, int n) {
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = THREADS / 64;
    // merges k = 2 .. 128: entirely chunk-local
    for (int c = wave * CHUNK; c < n2 && c < n; c += NW * CHUNK) {
        for (int k = 2; k <= CHUNK && k <= n2; k <<= 1) {
            const int hk = k >> 1;
            const int blk = lane / hk, l = lane - blk * hk;
            cmpx(a, c + blk * k + l, c + blk * k + (k - 1 - l), n);
            wave_lds_order();
            for (int j = k >> 2; j >= 1; j >>= 1) {
                const int b2 = lane / j, l2 = lane - b2 * j;
                const int lo = c + b2 * 2 * j + l2;
                cmpx(a, lo, lo + j, n);
                wave_lds_order();
            }
        }
    }
    __syncthreads();
    const int half = n2 >> 1;
    for (int k = 2 * CHUNK; k <= n2; k <<= 1) {
        const int hk = k >> 1;
        for (int i = threadIdx.x; i < half; i += THREADS) {  // mirror step: spans the whole k-block
            const int blk = i / hk, l = i - blk * hk;
            cmpx(a, blk * k + l, blk * k + (k - 1 - l), n);
        }
        __syncthreads();
        for (int j = k >> 2; j >= CHUNK; j >>= 1) {  // long-distance halving steps
            for (int i = threadIdx.x; i < half; i += THREADS) {
                const int blk = i / j, l = i - blk * j;
                const int lo = blk * 2 * j + l;
                cmpx(a, lo, lo + j, n);
            }
            __syncthreads();
        }
        local_halving_steps(a, n, n2, CHUNK / 2, wave, NW, lane);
        __syncthreads();
    }
}

// fallback for lists that do not fit in LDS: the same network in global memory, a barrier after every step
__device__ inline void bitonic_sort_global(uint64_t* a, int n, int nthreads) {
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    const int half = n2 >> 1;
    for (int k = 2; k <= n2; k <<= 1) {
        const int hk = k >> 1;
        for (int i = threadIdx.x; i < half; i += nthreads) {
            const int blk = i / hk, l = i - blk * hk;
            cmpx(a, blk * k + l, blk * k + (k - 1 - l), n);
        }
        __syncthreads();
        for (int j = k >> 2; j >= 1; j >>= 1) {
            for (int i = threadIdx.x; i < half; i += nthreads) {
                const int blk = i / j, l = i - blk * j;
                const int lo = blk * 2 * j + l;
                cmpx(a, lo, lo + j, n);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// long lists (2048 < n <= 16384): stable LSD radix sort on the depth bits, 8 bits per pass, 1024 threads, all in LDS.
//
// A bitonic network needs log2(n)^2 / 2 ~ 100 dependent steps for 16 k keys (169 us measured for one 15 k-entry
// list -- the critical path of the whole launch); a radix sort needs one pass per depth byte that actually varies.
// LDS holds the 32-bit depth keys (64 KiB) and two 16-bit permutations (2 x 32 KiB) that the passes ping-pong
// between; the flat ids stay in the list's global segment and are gathered once at the end.  Equal depth bits are
// rare; the stable passes leave such a run in the (arbitrary) input order, and a fix-up orders each run by flat id
// -- the order of upstream's stable sort on the full 64-bit key.
//
// One pass: wave w owns positions [1024 w, 1024 w + 1024) in 16 rounds of 64 consecutive entries.  In a round the
// lanes holding the same byte find each other with 8 ballots (a match-any emulation), their order inside the round is
// the lane order, and the wave's running count of that byte (LDS, wave-private, no atomics) gives the entry's rank
// among the wave's entries with that byte.  After a workgroup barrier the per-(byte, wave) counts are turned into
// global offsets (byte-major, then wave) and every entry is scattered to offset + rank: stable, as LSD requires.
// ---------------------------------------------------------------------------------------------------
constexpr int RADIX_WAVES = 16;
constexpr int RADIX_CAP = 16384;
constexpr int RADIX_MAX_RUN = 48;
struct RadixShared {
    uint32_t depth[RADIX_CAP];
    uint16_t perm[2][RADIX_CAP];
    int cnt[RADIX_WAVES][256];
    int flag;
};

// seg: the list's n 64-bit keys (depth bits << 32 | flat id) in global memory, any order.  On return
// sh.perm[result][i] is the position in seg of the i-th entry of the sorted list.
__device__ inline int radix_sort_long(const uint64_t* __restrict__ seg, int n, RadixShared& sh) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int i = threadIdx.x; i < n; i += 1024) {
        sh.depth[i] = (uint32_t)(seg[i] >> 32);
        sh.perm[0][i] = (uint16_t)i;
    }
    __syncthreads();
    int cur = 0;
    for (int pass = 0; pass < 4; ++pass) {
        for (int i = threadIdx.x; i < RADIX_WAVES * 256; i += 1024) (&sh.cnt[0][0])[i] = 0;
        if (threadIdx.x == 0) sh.flag = 0;
        __syncthreads();
        uint16_t item[16];
        uint8_t digit[16];
        int rank[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = wv * 1024 + r * 64 + lane;
            const bool valid = p < n;
            item[r] = valid ? sh.perm[cur][p] : (uint16_t)0;
            const int d = valid ? (int)((sh.depth[item[r]] >> (8 * pass)) & 255u) : 0;
            digit[r] = (uint8_t)d;
            uint64_t peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint64_t bm = __builtin_amdgcn_ballot_w64(((d >> b) & 1) != 0);
                peers &= ((d >> b) & 1) ? bm : ~bm;
            }
            rank[r] = 0;
            int before = 0;
            if (valid) before = sh.cnt[wv][d];  // every peer reads the same word before the leader updates it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                rank[r] = before + __builtin_popcountll(peers & lt);
                if ((peers & lt) == 0ull) sh.cnt[wv][d] = before + __builtin_popcountll(peers);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        __syncthreads();
        // byte totals -> offsets (byte-major, wave-minor); a byte holding all n entries makes the pass a no-op
        int total = 0;
        if (threadIdx.x < 256) {
#pragma unroll
            for (int w = 0; w < RADIX_WAVES; ++w) {
                const int c = sh.cnt[w][threadIdx.x];
                sh.cnt[w][threadIdx.x] = total;  // exclusive over the waves, inside this byte
                total += c;
            }
            if (total == n) sh.flag = 1;
        }
        int scan_total;
        const int incl = block_incl_scan_w<RADIX_WAVES>(threadIdx.x < 256 ? total : 0, &scan_total);
        __syncthreads();
        if (sh.flag == 0) {
            if (threadIdx.x < 256) {
                const int base = incl - total;
#pragma unroll
                for (int w = 0; w < RADIX_WAVES; ++w) sh.cnt[w][threadIdx.x] += base;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = wv * 1024 + r * 64 + lane;
                if (p < n) sh.perm[cur ^ 1][sh.cnt[wv][digit[r]] + rank[r]] = item[r];
            }
            cur ^= 1;
        }
        __syncthreads();
    }
    // runs of equal depth bits: order by flat id (the thread at the head of a run sorts it; runs are rare and
    // short).  A run longer than RADIX_MAX_RUN (many splats at exactly the same depth, e.g. a fronto-parallel plane)
    // would serialise in one thread: give up (-1) and let the caller sort the full 64-bit keys with the network.
    if (threadIdx.x == 0) sh.flag = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const uint32_t di = sh.depth[sh.perm[cur][i]];
        const bool head = (i == 0 || sh.depth[sh.perm[cur][i - 1]] != di) && i + 1 < n &&
                          sh.depth[sh.perm[cur][i + 1]] == di;
        if (head) {
            int e = i + 1;
            while (e < n && e - i <= RADIX_MAX_RUN && sh.depth[sh.perm[cur][e]] == di) ++e;
            if (e - i > RADIX_MAX_RUN) {
                sh.flag = 1;
                continue;
            }
            for (int a = i + 1; a < e; ++a) {  // insertion sort on the flat id (low 32 bits of the global key)
                const uint16_t pa = sh.perm[cur][a];
                const uint32_t ida = (uint32_t)seg[pa];
                int b = a - 1;
                while (b >= i && (uint32_t)seg[sh.perm[cur][b]] > ida) {
                    sh.perm[cur][b + 1] = sh.perm[cur][b];
                    --b;
                }
                sh.perm[cur][b + 1] = pa;
            }
        }
    }
    __syncthreads();
    return sh.flag ? -1 : cur;
}

// ---------------------------------------------------------------------------------------------------
// short lists (n <= 512 / 1024 / 2048, see tile_sort_short_kernel): one WAVE per tile, the bitonic network in registers.
//
// Element i of the (virtually +inf padded) list lives in lane i / EPL, register i % EPL.  Compare-exchange steps
// with distance < EPL stay inside a lane (register pairs); the others exchange whole registers with lane ^ (j / EPL)
// through the cross-lane network.  No LDS, no barriers, and four tiles per 256-thread workgroup instead of four waves
// that mostly wait at the barriers of one tile -- 53 -> ~25 us for the 5440 lists (~311 entries) of the benchmark.
// ---------------------------------------------------------------------------------------------------
// v of lane ^ d for d = 1 .. 32 WITHOUT the LDS crossbar (round 5): the network is a chain of ~21 dependent cross-lane
// steps per list and ds_bpermute answers after ~100+ cycles; DPP moves (d <= 8, inside a 16-lane row) and the gfx950
// permlane swaps (d = 16, 32) are VALU instructions with a few cycles of latency.
//   d = 1, 2: quad_perm;  d = 4: row_half_mirror (l -> 7 - l) then the quad reversed;  d = 8: row_ror:8;
//   d = 16 / 32: v_permlane16_swap / v_permlane32_swap of the register with itself leave the partner row / half in one of
//   the two results, picked by the lane's own bit.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t lane_xor32(uint32_t v, int d, int lane) {
    if (d == 1) return dpp_mov<0xB1>(v);
    if (d == 2) return dpp_mov<0x4E>(v);
    if (d == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));
    if (d == 8) return dpp_mov<0x128>(v);
    if (d == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    }
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane & 32) ? r[0] : r[1];
}
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v, int d, int lane) {
    return ((uint64_t)lane_xor32((uint32_t)(v >> 32), d, lane) << 32) | lane_xor32((uint32_t)v, d, lane);
}

template <int EPL>
__device__ __forceinline__ void wave_bitonic_sort(uint64_t (&key)[EPL], int lane) {
    constexpr int N2 = 64 * EPL;
#pragma unroll
    for (int k = 2; k <= N2; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j >= EPL) {
                const int d = j / EPL;                       // partner lane distance
                const bool lower = (lane & d) == 0;           // this lane holds the lower index of each pair
                const bool asc = k >= N2 ? true : ((lane & (k / EPL)) == 0);
                const bool want_min = lower == asc;
#pragma unroll
                for (int r = 0; r < EPL; ++r) {
                    const uint64_t other = lane_xor64(key[r], d, lane);
                    const bool other_less = other < key[r];
                    key[r] = (other_less == want_min) ? other : key[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < EPL; ++r) {
                    if ((r & j) == 0) {
                        // direction of the k-block this pair belongs to: a register bit for k < EPL, a lane bit else
                        const bool asc = k >= N2 ? true : (k < EPL ? ((r & k) == 0) : ((lane & (k / EPL)) == 0));
                        const uint64_t a = key[r], b = key[r | j];
                        const bool swap = (b < a) == asc;
                        key[r] = swap ? b : a;
                        key[r | j] = swap ? a : b;
                    }
                }
            }
        }
    }
}

template <int EPL>
__device__ __forceinline__ void sort_tile_in_wave(const uint64_t* __restrict__ seg, int n, int s, uint64_t hi_bits,
                                                  int32_t* __restrict__ flatten_ids, uint64_t* __restrict__ isect_ids,
                                                  int lane) {
    uint64_t key[EPL];
#pragma unroll
    for (int r = 0; r < EPL; ++r) {  // coalesced load; the initial arrangement is irrelevant to a sort
        const int i = r * 64 + lane;
        key[r] = i < n ? seg[i] : ~0ull;
    }
    wave_bitonic_sort<EPL>(key, lane);
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const int i = lane * EPL + r;
        if (i < n) {
            flatten_ids[s + i] = (int32_t)(uint32_t)key[r];
            if (isect_ids) isect_ids[s + i] = hi_bits | (key[r] >> 32);
        }
    }
}

// lists of up to 64 * MAXEPL entries sort in registers; three builds of the kernel (MAXEPL = 8 / 16 / 32 keys per lane:
// 512 / 1024 / 2048 entries) because the register count of the longest network sets the occupancy of all of them
// (MAXEPL 16: 76 VGPRs, 32: 138 VGPRs = 3 waves per SIMD)
constexpr int SHORT_SORT_LDS_KEYS = 2048;  // 16 KiB: longer lists belong to the long-list launch (or sort in global memory)

// A workgroup sorts one list of any length: bitonic network in LDS when it fits, in place in global memory else.
template <int THREADS>
__device__ __forceinline__ void sort_tile_by_block(uint64_t* lds_keys, int lds_cap, uint64_t* seg, int n, int s,
                                                   uint64_t hi_bits, int32_t* __restrict__ flatten_ids,
                                                   uint64_t* __restrict__ isect_ids) {
    if (n <= lds_cap) {
        for (int i = threadIdx.x; i < n; i += THREADS) lds_keys[i] = seg[i];
        __syncthreads();
        if (n > 1) bitonic_sort_lds<THREADS>(lds_keys, n);
        for (int i = threadIdx.x; i < n; i += THREADS) {
            const uint64_t k = lds_keys[i];
            flatten_ids[s + i] = (int32_t)(uint32_t)k;
            if (isect_ids) isect_ids[s + i] = hi_bits | (k >> 32);
        }
    } else {
        // pathological tile: sort in place in global memory (same workgroup, barrier-ordered)
        bitonic_sort_global(seg, n, THREADS);
        for (int i = threadIdx.x; i < n; i += THREADS) {
            const uint64_t k = seg[i];
            flatten_ids[s + i] = (int32_t)(uint32_t)k;
            if (isect_ids) isect_ids[s + i] = hi_bits | (k >> 32);
        }
    }
}

// All lists of up to n_max entries in ONE launch: four tiles per workgroup; each wave sorts its tile in registers
// when the list has <= 64 * MAXEPL entries (nearly all of them), and the workgroup then takes the longer ones among
// its four
// tiles together, one after the other (LDS network).
// MAXEPL is chosen from the longest list expected: longer lists than it covers go one at a time through the
// workgroup's LDS network, which is several times slower per list (800 k splats, mean list 810: 105 us with MAXEPL 8,
// 56 us with 16; 1.5 M splats, mean 1500: 226 us with 16, 129 us with 32), while a build wider than needed costs
// occupancy (300 k splats: 27.2 / 28.7 / 28.9 us with 8 / 16 / 32).
template <int MAXEPL>
__global__ void __launch_bounds__(256) tile_sort_short_kernel(int n_tiles_total, int tile_bits,
                                                                const int32_t* __restrict__ tile_offsets,
                                                                uint64_t* __restrict__ sort_keys,
                                                                int32_t* __restrict__ flatten_ids,
                                                                uint64_t* __restrict__ isect_ids, int tiles_per_cam,
                                                                int n_max) {
    __shared__ __attribute__((aligned(16))) uint64_t lds_keys[SHORT_SORT_LDS_KEYS];
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    int s = 0, n = 0;
    if (t < n_tiles_total) {
        s = tile_offsets[t];
        n = tile_offsets[t + 1] - s;
    }
    constexpr int WAVE_MAX = 64 * MAXEPL;
    if (n > 0 && n <= WAVE_MAX) {
        const uint64_t* seg = sort_keys + s;
        const int cam = t / tiles_per_cam, tl = t - cam * tiles_per_cam;
        const uint64_t hi_bits = (((uint64_t)cam << tile_bits) | (uint64_t)tl) << 32;
        if (n <= 128)
            sort_tile_in_wave<2>(seg, n, s, hi_bits, flatten_ids, isect_ids, lane);
        else if (n <= 256)
            sort_tile_in_wave<4>(seg, n, s, hi_bits, flatten_ids, isect_ids, lane);
        else if (MAXEPL == 8 || n <= 512)
            sort_tile_in_wave<8>(seg, n, s, hi_bits, flatten_ids, isect_ids, lane);
        else if (MAXEPL == 16 || n <= 1024)
            sort_tile_in_wave<16>(seg, n, s, hi_bits, flatten_ids, isect_ids, lane);
        else
            sort_tile_in_wave<32>(seg, n, s, hi_bits, flatten_ids, isect_ids, lane);
    }
    if (!__syncthreads_or(n > WAVE_MAX && n <= n_max)) return;
    for (int w = 0; w < 4; ++w) {
        const int t2 = blockIdx.x * 4 + w;
        if (t2 >= n_tiles_total) break;
        const int s2 = tile_offsets[t2], n2 = tile_offsets[t2 + 1] - s2;
        if (n2 <= WAVE_MAX || n2 > n_max) continue;
        const int cam = t2 / tiles_per_cam, tl = t2 - cam * tiles_per_cam;
        const uint64_t hi_bits = (((uint64_t)cam << tile_bits) | (uint64_t)tl) << 32;
        sort_tile_by_block<256>(lds_keys, SHORT_SORT_LDS_KEYS, sort_keys + s2, n2, s2, hi_bits, flatten_ids, isect_ids);
        __syncthreads();  // lds_keys is reused
    }
}

// tiles whose list is longer than `min_len`, in no particular order: long_ids[0 .. *long_count)
__global__ void __launch_bounds__(256) long_lists_kernel(int nt, const int32_t* __restrict__ tile_offsets, int min_len,
                                                           int32_t* __restrict__ long_ids,
                                                           int32_t* __restrict__ long_count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nt && tile_offsets[t + 1] - tile_offsets[t] > min_len) long_ids[atomicAdd(long_count, 1)] = t;
}

// Lists with n_min <= length <= n_max are sorted by this launch (the others belong to the launch of the other
// variant).  long_ids != NULL: instead of one workgroup per tile, a small grid walks that list of tiles -- the
// long-list variant needs 128 KiB of LDS per workgroup, and dispatching one such workgroup per tile just to find
// that the tile is short costs more than the sort.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) tile_sort_kernel(int n_tiles_total, int lds_cap, int tile_bits,
                                                              const int32_t* __restrict__ tile_offsets,
                                                              uint64_t* __restrict__ sort_keys,
                                                              int32_t* __restrict__ flatten_ids,
                                                              uint64_t* __restrict__ isect_ids, int tiles_per_cam,
                                                              int n_min, int n_max,
                                                              const int32_t* __restrict__ long_ids,
                                                              const int32_t* __restrict__ long_count) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_keys[];
    for (int h = blockIdx.x;; h += gridDim.x) {
        int t = h;
        if (long_ids) {
            if (h >= *long_count) return;
            t = long_ids[h];
        } else if (t >= n_tiles_total) {
            return;
        }
        const int s = tile_offsets[t], e = tile_offsets[t + 1];
        const int n = e - s;
        if (n > 0 && n >= n_min && n <= n_max) {
            uint64_t* seg = sort_keys + s;
            const int cam = t / tiles_per_cam, tl = t - cam * tiles_per_cam;
            const uint64_t hi_bits = (((uint64_t)cam << tile_bits) | (uint64_t)tl) << 32;
            if (n <= lds_cap) {
                if constexpr (THREADS == 64 * RADIX_WAVES) {
                    RadixShared& rs = *reinterpret_cast<RadixShared*>(lds_keys);
                    const int cur = radix_sort_long(seg, n, rs);
                    if (cur >= 0) {
                        for (int i = threadIdx.x; i < n; i += THREADS) {
                            const uint64_t k = seg[rs.perm[cur][i]];
                            flatten_ids[s + i] = (int32_t)(uint32_t)k;
                            if (isect_ids) isect_ids[s + i] = hi_bits | (k >> 32);
                        }
                    } else {  // long runs of equal depth: the network on the full keys (128 KiB of the same LDS)
                        __syncthreads();
                        for (int i = threadIdx.x; i < n; i += THREADS) lds_keys[i] = seg[i];
                        __syncthreads();
                        bitonic_sort_lds<THREADS>(lds_keys, n);
                        for (int i = threadIdx.x; i < n; i += THREADS) {
                            const uint64_t k = lds_keys[i];
                            flatten_ids[s + i] = (int32_t)(uint32_t)k;
                            if (isect_ids) isect_ids[s + i] = hi_bits | (k >> 32);
                        }
                    }
                } else {
                    for (int i = threadIdx.x; i < n; i += THREADS) lds_keys[i] = seg[i];
                    __syncthreads();
                    if (n > 1) bitonic_sort_lds<THREADS>(lds_keys, n);
                    for (int i = threadIdx.x; i < n; i += THREADS) {
                        const uint64_t k = lds_keys[i];
                        flatten_ids[s + i] = (int32_t)(uint32_t)k;
                        if (isect_ids) isect_ids[s + i] = hi_bits | (k >> 32);
                    }
                }
            } else {
                // pathological tile: sort in place in global memory (same workgroup, barrier-ordered)
                bitonic_sort_global(seg, n, THREADS);
                for (int i = threadIdx.x; i < n; i += THREADS) {
                    const uint64_t k = seg[i];
                    flatten_ids[s + i] = (int32_t)(uint32_t)k;
                    if (isect_ids) isect_ids[s + i] = hi_bits | (k >> 32);
                }
            }
        }
        if (!long_ids) return;
        __syncthreads();  // lds_keys is reused by the next long list
    }
}

// ---------------------------------------------------------------------------------------------------
// huge lists (n > 16384; round 3): sorted chunks + parallel merge passes
//
// A list that does not fit the LDS radix sort used to be sorted in place in global memory by ONE workgroup (a barrier
// after each of ~150 network steps): 7.8 ms for a 197 k-entry list, 67 % of the whole step on a scene with 70 % of the
// splats in 2 % of the screen (scripts/heavy_tail.py 0.7 0.02).  Now every 16384-entry chunk of such a list is radix-
// sorted in LDS by its own workgroup (the keys written back in place), and log2(#chunks) merge passes double the run
// length: a pass cuts every list into 4096-key output blocks, a workgroup finds its block's two input ranges with a
// merge-path search (two binary searches over global memory), stages them in LDS and its 256 threads merge 16 keys
// each (their own merge-path split in LDS).  The 64-bit keys are unique (flat id in the low half), so the merge needs no
// tie rule.  Passes ping-pong between the key arena and a scratch region that is dead after emit (owner / tile / rank
// triples: 12 bytes per bounding-box intersection); the number of passes a frame needs is only known on the device
// (a word holding the largest chunk count), so a fixed number is launched and the surplus ones return at once.
// ---------------------------------------------------------------------------------------------------
constexpr int MERGE_BLOCK = 4096;

// number of elements the first d outputs of merge(A, B) take from A (keys unique)
__device__ inline int merge_path(const uint64_t* __restrict__ A, int lenA, const uint64_t* __restrict__ B, int lenB, int d) {
    int lo = max(0, d - lenB), hi = min(d, lenA);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (A[mid] < B[d - 1 - mid]) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(1024) huge_chunk_sort_kernel(const int32_t* __restrict__ tile_offsets,
                                                               uint64_t* __restrict__ sort_keys,
                                                               const int32_t* __restrict__ long_ids,
                                                               const int32_t* __restrict__ long_count,
                                                               int32_t* __restrict__ max_chunks) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_keys[];
    RadixShared& rs = *reinterpret_cast<RadixShared*>(lds_keys);
    const int n_long = *long_count;
    int base = 0;  // chunks of the huge lists before this one: work is dealt over ALL lists' chunks, not per list
    for (int h = 0; h < n_long; ++h) {
        const int t = long_ids[h];
        const int s = tile_offsets[t], n = tile_offsets[t + 1] - s;
        if (n <= RADIX_CAP) continue;
        const int nch = (n + RADIX_CAP - 1) / RADIX_CAP;
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(max_chunks, nch);
        const int first = ((int)blockIdx.x - base % (int)gridDim.x + (int)gridDim.x) % (int)gridDim.x;
        base += nch;
        for (int c = first; c < nch; c += gridDim.x) {
            uint64_t* seg = sort_keys + s + (size_t)c * RADIX_CAP;
            const int m = min(RADIX_CAP, n - c * RADIX_CAP);
            const int cur = radix_sort_long(seg, m, rs);
            if (cur >= 0) {
                uint64_t k[RADIX_CAP / 1024];
#pragma unroll
                for (int r = 0; r < RADIX_CAP / 1024; ++r) {
                    const int i = threadIdx.x + 1024 * r;
                    k[r] = i < m ? seg[rs.perm[cur][i]] : 0ull;
                }
                __syncthreads();  // every gather done before the segment is overwritten
#pragma unroll
                for (int r = 0; r < RADIX_CAP / 1024; ++r) {
                    const int i = threadIdx.x + 1024 * r;
                    if (i < m) seg[i] = k[r];
                }
            } else {  // long runs of equal depth: the network on the full keys
                __syncthreads();
                for (int i = threadIdx.x; i < m; i += 1024) lds_keys[i] = seg[i];
                __syncthreads();
                bitonic_sort_lds<1024>(lds_keys, m);
                for (int i = threadIdx.x; i < m; i += 1024) seg[i] = lds_keys[i];
            }
            __syncthreads();  // the LDS state is reused by the next chunk
        }
    }
}

__global__ void __launch_bounds__(256) huge_merge_kernel(int pass, const int32_t* __restrict__ tile_offsets,
                                                         uint64_t* __restrict__ keys, uint64_t* __restrict__ tmp,
                                                         const int32_t* __restrict__ long_ids,
                                                         const int32_t* __restrict__ long_count,
                                                         const int32_t* __restrict__ max_chunks) {
    if ((1 << pass) >= *max_chunks) return;  // every huge list is one sorted run already
    const uint64_t* src = (pass & 1) ? tmp : keys;
    uint64_t* dst = (pass & 1) ? keys : tmp;
    __shared__ uint64_t buf[MERGE_BLOCK];
    __shared__ int sp[2];
    const int R = RADIX_CAP << pass;  // run length entering this pass
    const int n_long = *long_count;
    int base = 0;  // output blocks of the huge lists before this one (work dealt over all of them)
    for (int h = 0; h < n_long; ++h) {
        const int t = long_ids[h];
        const int s = tile_offsets[t], n = tile_offsets[t + 1] - s;
        if (n <= RADIX_CAP) continue;
        const int nblk = (n + MERGE_BLOCK - 1) / MERGE_BLOCK;
        const int first = ((int)blockIdx.x - base % (int)gridDim.x + (int)gridDim.x) % (int)gridDim.x;
        base += nblk;
        for (int b = first; b < nblk; b += gridDim.x) {
            const int o0 = b * MERGE_BLOCK, o1 = min(n, o0 + MERGE_BLOCK), cnt = o1 - o0;
            const int a0 = (o0 / (2 * R)) * (2 * R);  // MERGE_BLOCK divides R: a block never straddles two pairs
            const int a1 = min(n, a0 + R), b1 = min(n, a0 + 2 * R);
            const int lenA = a1 - a0, lenB = b1 - a1;
            const uint64_t* A = src + s + a0;
            const uint64_t* B = src + s + a1;
            if (threadIdx.x < 2) sp[threadIdx.x] = merge_path(A, lenA, B, lenB, (threadIdx.x ? o1 : o0) - a0);
            __syncthreads();
            const int ia0 = sp[0], ia1 = sp[1];
            const int ib0 = (o0 - a0) - ia0;
            const int na = ia1 - ia0, nb = cnt - na;
            for (int i = threadIdx.x; i < cnt; i += 256) buf[i] = i < na ? A[ia0 + i] : B[ib0 + (i - na)];
            __syncthreads();
            const int dd = 16 * threadIdx.x;
            if (dd < cnt) {
                int ja = merge_path(buf, na, buf + na, nb, dd), jb = dd - ja;
                uint64_t* out = dst + s + o0 + dd;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (dd + k < cnt) {
                        const bool take_a = jb >= nb || (ja < na && buf[ja] < buf[na + jb]);
                        out[k] = take_a ? buf[ja++] : buf[na + jb++];
                    }
                }
            }
            __syncthreads();  // buf / sp are reused by the next block
        }
    }
}

__global__ void __launch_bounds__(256) huge_finish_kernel(int tile_bits, int tiles_per_cam,
                                                          const int32_t* __restrict__ tile_offsets,
                                                          const uint64_t* __restrict__ keys,
                                                          const uint64_t* __restrict__ tmp,
                                                          int32_t* __restrict__ flatten_ids,
                                                          uint64_t* __restrict__ isect_ids,
                                                          const int32_t* __restrict__ long_ids,
                                                          const int32_t* __restrict__ long_count,
                                                          const int32_t* __restrict__ max_chunks) {
    int passes = 0;
    while ((1 << passes) < *max_chunks) ++passes;
    const uint64_t* src = (passes & 1) ? tmp : keys;  // where the last pass left the lists
    const int n_long = *long_count;
    for (int h = 0; h < n_long; ++h) {
        const int t = long_ids[h];
        const int s = tile_offsets[t], n = tile_offsets[t + 1] - s;
        if (n <= RADIX_CAP) continue;
        const int cam = t / tiles_per_cam, tl = t - cam * tiles_per_cam;
        const uint64_t hi_bits = (((uint64_t)cam << tile_bits) | (uint64_t)tl) << 32;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
            const uint64_t k = src[s + i];
            flatten_ids[s + i] = (int32_t)(uint32_t)k;
            if (isect_ids) isect_ids[s + i] = hi_bits | (k >> 32);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// fused single-pass lists (round 5): offsets / schedule / counts behind bin_kernel<.., true>, and the sort that reads
// the strided key segments
// ---------------------------------------------------------------------------------------------------
// Where the i-th key of a tile lives in its strided segment [copy][seg_stride]: the copies' fill counts are
// wave-uniform, entry i of the concatenation sits (i - prefix_c) into copy c.
// ---- offsets / schedule / counts of the fused path -----------------------------------------------------------------------
// tile_scan_kernel without the per-(copy, tile) base table, and with every global load of a phase in flight before the
// first barrier (tile_scan_kernel loops "8 loads -> block scan" six times for 5440 tiles: six dependent trips to
// counters the atomics left at the memory side, 12.9 us; here the lengths go to LDS in one sweep: 9.7 us).
//   workgroup 0: list lengths (sum of the counter copies) -> tile_offsets, longest list, the counts for the host;
//                EMPTY lists when an arena or a key segment was too small (seg_stride)
//   workgroup 1: chunk totals of keep_scan -> chunk bases; stats[1]
//   workgroup 2: the tile schedule (as tile_scan_kernel)
// (Tried and dropped, round 5: folding this launch into the sort launch.  (a) A decoupled look-back among the 1360 sort
// workgroups: they all start at once, so prefixes propagate one 64-group window per round trip -- 37 us against 25 + 9.7;
// (b) an auxiliary workgroup of the sort launch scanning the lengths while the others sort, positions handed over through
// {valid, value} words: correct, but a 256-thread workgroup needs 32 dependent rounds of loads for 5440 x 8 counters and
// the sort waves wait for it -- 40 - 55 us.)
constexpr int FIN_LDS_TILES = 8192;  // lengths kept in LDS per sweep (32 KiB)
__global__ void __launch_bounds__(TSCAN_THREADS) tile_finish_kernel(int nt, const int32_t* __restrict__ tile_count,
                                                                      int cstride, int32_t* __restrict__ tile_offsets,
                                                                      int64_t* __restrict__ stats,
                                                                      int32_t* __restrict__ tile_order,
                                                                      int64_t capacity_box, int64_t capacity_listed,
                                                                      int seg_stride, int32_t* __restrict__ keep_scan,
                                                                      int n_chunks, int heavy_len,
                                                                      int64_t* stats_mirror, int64_t stats_seq) {
    __shared__ int s_len[FIN_LDS_TILES];
    __shared__ int smax[TSCAN_THREADS / 64];
    __shared__ int hist[ORDER_BUCKETS];
    if (blockIdx.x == 1) {
        // chunk totals -> bases: all of a thread's words requested before the first scan
        constexpr int PER = 4;
        int base = 0;
        for (int c0 = 0; c0 < n_chunks; c0 += TSCAN_THREADS * PER) {
            int v[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int c = c0 + threadIdx.x * PER + k;
                v[k] = (c < n_chunks) ? keep_scan[(size_t)c * (KEEP_CHUNK + 1)] : 0;
            }
            int sum = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k) sum += v[k];
            int total;
            const int inc = block_incl_scan_w<TSCAN_THREADS / 64>(sum, &total);
            int run = base + inc - sum;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int c = c0 + threadIdx.x * PER + k;
                if (c < n_chunks) keep_scan[(size_t)c * (KEEP_CHUNK + 1)] = run;
                run += v[k];
            }
            base += total;
        }
        if (threadIdx.x == 0) stats[1] = (int64_t)base;
        return;
    }
    const bool order_wg = blockIdx.x == 2;
    auto length_of = [&](int i) { return i < FIN_LDS_TILES ? s_len[i] : tile_total(tile_count, cstride, i); };
    int carry = 0, mx = 0;
    constexpr int TPT = FIN_LDS_TILES / TSCAN_THREADS;  // consecutive tiles per thread in the scan phase
    for (int sweep = 0; sweep < nt; sweep += FIN_LDS_TILES) {
        // phase 1: coalesced loads of the counter copies, every one of them independent
        int v[TPT];
#pragma unroll
        for (int m = 0; m < TPT; ++m) {
            const int i = sweep + m * TSCAN_THREADS + threadIdx.x;
            int sum = 0;
            if (i < nt) {
#pragma unroll
                for (int c = 0; c < TC_COPIES; ++c) sum += tile_count[(size_t)c * cstride + i];
            }
            v[m] = sum;
        }
        if (sweep > 0) __syncthreads();  // the previous sweep's readers are done with s_len
#pragma unroll
        for (int m = 0; m < TPT; ++m) {
            s_len[m * TSCAN_THREADS + threadIdx.x] = v[m];
            mx = max(mx, v[m]);
        }
        __syncthreads();
        if (order_wg) {
            if (nt > FIN_LDS_TILES) continue;  // (only the maximum is needed from the later sweeps; lengths re-read)
            break;
        }
        // phase 2: thread t scans tiles [TPT t, TPT t + TPT) of the sweep
        int mine[TPT], sum = 0;
#pragma unroll
        for (int m = 0; m < TPT; ++m) {
            mine[m] = s_len[threadIdx.x * TPT + m];
            sum += mine[m];
        }
        int total;
        const int inc = block_incl_scan_w<TSCAN_THREADS / 64>(sum, &total);
        int run = carry + inc - sum;
#pragma unroll
        for (int m = 0; m < TPT; ++m) {
            const int i = sweep + threadIdx.x * TPT + m;
            if (i < nt) tile_offsets[i] = run;
            run += mine[m];
        }
        carry += total;
    }
    if (order_wg && nt > FIN_LDS_TILES) {  // s_len must hold the FIRST sweep again for length_of()
        __syncthreads();
        for (int i = threadIdx.x; i < FIN_LDS_TILES; i += TSCAN_THREADS) s_len[i] = tile_total(tile_count, cstride, i);
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mx;
    __syncthreads();
    int longest = 0;
#pragma unroll
    for (int k = 0; k < TSCAN_THREADS / 64; ++k) longest = max(longest, smax[k]);
    if (!order_wg) {
        if (threadIdx.x == 0) {
            tile_offsets[nt] = carry;
            stats[2] = (int64_t)longest;
            // the host's copy of {I_box, I_listed, longest list}: written straight into its pinned, device-mapped slot, the
            // sequence number last (the host polls that word: no event, no marker packet in the queue)
            if (stats_mirror) {
                stats_mirror[0] = stats[0];
                stats_mirror[1] = (int64_t)carry;
                stats_mirror[2] = (int64_t)longest;
                __threadfence_system();
                if (stats_seq)
                    __hip_atomic_store(&stats_mirror[3], stats_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        // an arena or a key segment too small: EMPTY lists for every consumer already enqueued; the counts stay true
        if (stats[0] > capacity_box || (int64_t)carry > capacity_listed || longest > seg_stride) {
            __syncthreads();
            for (int i = threadIdx.x; i <= nt; i += TSCAN_THREADS) tile_offsets[i] = 0;
        }
        return;
    }
    if (!tile_order) return;
    auto bucket = [&](int len) {
        const int q = longest > 0 ? (int)(((int64_t)len * (ORDER_BUCKETS - 1)) / longest) : 0;
        return ORDER_BUCKETS - 1 - q;  // bucket 0 = the longest lists
    };
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += TSCAN_THREADS) atomicAdd(&hist[bucket(length_of(i))], 1);
    __syncthreads();
    {
        const int mine = hist[threadIdx.x];
        int total;
        const int inc = block_incl_scan_w<TSCAN_THREADS / 64>(mine, &total);
        hist[threadIdx.x] = inc - mine;
    }
    __syncthreads();
    // heavy set: as tile_scan_kernel (a function of the tiles' length classes only -> deterministic)
    __shared__ int s_heavy, s_cut;
    if (threadIdx.x == 0) s_cut = 0;
    __syncthreads();
    if (heavy_len > 0 && longest >= heavy_len) {
        const int b_thr = bucket(heavy_len);
        if ((int)threadIdx.x <= b_thr && hist[threadIdx.x] <= (int)sched_max_heavy((size_t)nt))
            atomicMax(&s_cut, (int)threadIdx.x);
    }
    __syncthreads();
    if (threadIdx.x == 0) s_heavy = hist[s_cut];
    const int n_slots = (int)sched_slots((size_t)nt);
    for (int i = threadIdx.x; i < n_slots; i += TSCAN_THREADS) tile_order[i] = -1;
    __syncthreads();
    const int n_heavy = s_heavy;
    for (int i = threadIdx.x; i < nt; i += TSCAN_THREADS) {
        const int pos = atomicAdd(&hist[bucket(length_of(i))], 1);
        if (pos < n_heavy) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tile_order[4 * pos + q] = i | SCHED_HEAVY;
        } else {
            tile_order[3 * n_heavy + pos] = i;
        }
    }
}

// Where the i-th key of a tile lives in its strided segment [copy][seg_stride]: the copies' fill counts are
// wave-uniform, entry i of the concatenation sits (i - prefix_c) into copy c.
struct SegMap {
    int gap[TC_COPIES];   // seg_stride - count of copy c: what an index gains when it steps over the end of copy c
    int pre[TC_COPIES];   // entries in copies 0 .. c (inclusive prefix)
    int n;
    __device__ __forceinline__ void load(const int32_t* __restrict__ tile_count, int cstride, int t, int seg_stride) {
        int run = 0;
#pragma unroll
        for (int c = 0; c < TC_COPIES; ++c) {
            const int cnt = __builtin_amdgcn_readfirstlane(tile_count[(size_t)c * cstride + t]);
            run += cnt;
            pre[c] = run;
            gap[c] = seg_stride - cnt;
        }
        n = run;
    }
    __device__ __forceinline__ int offset_of(int i) const {
        int o = i;
#pragma unroll
        for (int c = 0; c + 1 < TC_COPIES; ++c) o += (i >= pre[c]) ? gap[c] : 0;
        return o;
    }
};

template <int EPL>
__device__ __forceinline__ void sort_seg_in_wave(const uint64_t* __restrict__ seg, const SegMap& sm, int s,
                                                 uint64_t hi_bits, int32_t* __restrict__ flatten_ids,
                                                 uint64_t* __restrict__ isect_ids, int lane) {
    uint64_t key[EPL];
    const int n = sm.n;
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const int i = r * 64 + lane;
        key[r] = i < n ? seg[sm.offset_of(i)] : ~0ull;
    }
    wave_bitonic_sort<EPL>(key, lane);
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
        const int i = lane * EPL + r;
        if (i < n) {
            flatten_ids[s + i] = (int32_t)(uint32_t)key[r];
            if (isect_ids) isect_ids[s + i] = hi_bits | (key[r] >> 32);
        }
    }
}

// tile_sort_short_kernel over the strided segments: ONE launch sorts every list (the fused path is only taken while
// the longest list expected fits SHORT_SORT_LDS_KEYS); output is PACKED at tile_offsets, exactly what the two-pass
// path writes, so no consumer can tell the difference.
template <int MAXEPL>
__global__ void __launch_bounds__(256) tile_sort_seg_kernel(int n_tiles_total, int tile_bits,
                                                              const int32_t* __restrict__ tile_offsets,
                                                              const int32_t* __restrict__ tile_count, int cstride,
                                                              const uint64_t* __restrict__ seg_keys, int seg_stride,
                                                              int32_t* __restrict__ flatten_ids,
                                                              uint64_t* __restrict__ isect_ids, int tiles_per_cam) {
    __shared__ __attribute__((aligned(16))) uint64_t lds_keys[SHORT_SORT_LDS_KEYS];
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    int s = 0, n = 0;
    SegMap sm;
    sm.n = 0;
    if (t < n_tiles_total) {
        sm.load(tile_count, cstride, t, seg_stride);
        s = __builtin_amdgcn_readfirstlane(tile_offsets[t]);
        // overflow (tile_finish_kernel emptied the lists): the packed range disagrees with the counters -> nothing to do
        n = (__builtin_amdgcn_readfirstlane(tile_offsets[t + 1]) - s == sm.n) ? sm.n : 0;
    }
    constexpr int WAVE_MAX = 64 * MAXEPL;
    if (n > 0 && n <= WAVE_MAX) {
        const uint64_t* seg = seg_keys + (size_t)t * TC_COPIES * (size_t)seg_stride;
        const int cam = t / tiles_per_cam, tl = t - cam * tiles_per_cam;
        const uint64_t hi_bits = (((uint64_t)cam << tile_bits) | (uint64_t)tl) << 32;
        if (n <= 128)
            sort_seg_in_wave<2>(seg, sm, s, hi_bits, flatten_ids, isect_ids, lane);
        else if (n <= 256)
            sort_seg_in_wave<4>(seg, sm, s, hi_bits, flatten_ids, isect_ids, lane);
        else if (MAXEPL == 8 || n <= 512)
            sort_seg_in_wave<8>(seg, sm, s, hi_bits, flatten_ids, isect_ids, lane);
        else if (MAXEPL == 16 || n <= 1024)
            sort_seg_in_wave<16>(seg, sm, s, hi_bits, flatten_ids, isect_ids, lane);
        else
            sort_seg_in_wave<32>(seg, sm, s, hi_bits, flatten_ids, isect_ids, lane);
    }
    if (!__syncthreads_or(n > WAVE_MAX)) return;
    for (int w = 0; w < 4; ++w) {  // the few lists beyond the register network: the workgroup's LDS network, one by one
        const int t2 = blockIdx.x * 4 + w;
        if (t2 >= n_tiles_total) break;
        SegMap m2;
        m2.load(tile_count, cstride, t2, seg_stride);
        const int s2 = tile_offsets[t2];
        const int n2 = (tile_offsets[t2 + 1] - s2 == m2.n) ? m2.n : 0;
        if (n2 <= WAVE_MAX || n2 > SHORT_SORT_LDS_KEYS) continue;  // (n2 <= seg_stride <= SHORT_SORT_LDS_KEYS by construction)
        const uint64_t* seg = seg_keys + (size_t)t2 * TC_COPIES * (size_t)seg_stride;
        const int cam = t2 / tiles_per_cam, tl = t2 - cam * tiles_per_cam;
        const uint64_t hi_bits = (((uint64_t)cam << tile_bits) | (uint64_t)tl) << 32;
        for (int i = threadIdx.x; i < n2; i += 256) lds_keys[i] = seg[m2.offset_of(i)];
        __syncthreads();
        bitonic_sort_lds<256>(lds_keys, n2);
        for (int i = threadIdx.x; i < n2; i += 256) {
            const uint64_t k = lds_keys[i];
            flatten_ids[s2 + i] = (int32_t)(uint32_t)k;
            if (isect_ids) isect_ids[s2 + i] = hi_bits | (k >> 32);
        }
        __syncthreads();  // lds_keys is reused
    }
}

}  // namespace mobgs

using namespace mobgs;

// MobgsTuning.heavy_tile_len: list length from which a tile is composited by a whole workgroup (scheduling policy,
// see tile_scan_kernel); MobgsTuning.longest_list_hint: longest list the caller expects (no longer consulted here).
// Both travel with the call -- the library keeps no mutable state.
constexpr int DENSE_MAX_TILES = 8192;  // 32 KiB of LDS

extern "C" {

// Layout of the scratch buffer shared by mobgs_isect_offsets and mobgs_isect_emit_sort (int32 units):
//   [tile_count TC_COPIES * nt | ticket (+3 pad) | status 2*(nb1+1)]  <- zeroed by one memset
//   [owner cap | tile cap | rank cap | chunk_cnt (cap >> 11) + 1]
static inline size_t count_stride(size_t n_tiles) { return (n_tiles * TC_STRIDE + 31) & ~(size_t)31; }
struct IsectScratch {
    int32_t *tile_count, *tickets, *chunk_cnt, *owner, *tile_of_j, *rank_of_j, *chunk_owner, *tile_base, *cum_enum;
    int owner_slots;
    uint64_t* status1;
    size_t zeroed_ints, total_ints;
    int nb1;
    IsectScratch(void* scratch, size_t n_gauss, size_t n_tiles, size_t capacity) {
        nb1 = (int)((n_gauss + SCAN_BLOCK - 1) / SCAN_BLOCK);
        // every counter copy of the fused path starts on its own 128-byte line; the two-pass path packs its copies at the start
        const size_t nt_pad = count_stride(n_tiles) * TC_COPIES;  // (even: the 64-bit status words stay 8-byte aligned)
        int32_t* p = (int32_t*)scratch;
        tile_count = p;
        tickets = p + nt_pad;
        status1 = (uint64_t*)(p + nt_pad + 4);
        zeroed_ints = nt_pad + 4 + 2 * (size_t)(nb1 + 1);
        owner = p + zeroed_ints;
        tile_of_j = owner + capacity;
        rank_of_j = tile_of_j + capacity;
        chunk_cnt = rank_of_j + capacity;
        owner_slots = (int)(capacity >> KEEP_CHUNK_LOG2) + 2;
        chunk_owner = chunk_cnt + (capacity >> KEEP_CHUNK_LOG2) + 1;
        tile_base = chunk_owner + owner_slots;
        cum_enum = tile_base + n_tiles * TC_COPIES;  // [n_gauss + 1]: the scan in the caller's enumeration order (fused path)
        total_ints = zeroed_ints + 3 * capacity + (capacity >> KEEP_CHUNK_LOG2) + 1 + (size_t)owner_slots +
                     n_tiles * TC_COPIES + n_gauss + 1;
    }
};

size_t mobgs_tile_order_len(int n_tiles) { return sched_slots((size_t)n_tiles); }

size_t mobgs_keep_scan_len(int capacity) { return keep_scan_len((size_t)capacity); }

size_t mobgs_isect_scratch_bytes(int n_gauss, int n_tiles, int capacity) {
    const IsectScratch L(nullptr, (size_t)n_gauss, (size_t)n_tiles, (size_t)capacity);
    return sizeof(int32_t) * (L.total_ints + 32);
}

int mobgs_isect_offsets(int C, int N, int tile_w, int tile_h, int width, int height, int cull, int capacity,
                        const int32_t* tiles_per_gauss, const float* means2d, const int32_t* radii,
                        const float* conics, const float* opacities, int opac_per_camera, int32_t* cum_tiles,
                        int32_t* keep_scan, int32_t* tile_offsets, int32_t* tile_order, int64_t capacity_listed,
                        int64_t* stats, void* scratch, const MobgsTuning* tuning, void* stream) {
    return mobgs::isect_offsets_launch(C, N, tile_w, tile_h, width, height, cull, capacity, tiles_per_gauss, means2d, radii,
                                       conics, opacities, opac_per_camera, cum_tiles, keep_scan, tile_offsets, tile_order,
                                       capacity_listed, stats, scratch, /*scratch_zeroed=*/false, /*stats_mirror=*/nullptr,
                                       /*stats_seq=*/0, tuning, stream);
}

}  // extern "C"

void mobgs::isect_zeroed_region(void* scratch, size_t n_gauss, size_t n_tiles, size_t capacity, int32_t** ptr,
                                size_t* count) {
    const IsectScratch L(scratch, n_gauss, n_tiles, capacity);
    *ptr = L.tile_count;
    *count = L.zeroed_ints;
}

int mobgs::isect_offsets_launch(int C, int N, int tile_w, int tile_h, int width, int height, int cull, int capacity,
                                const int32_t* tiles_per_gauss, const float* means2d, const int32_t* radii,
                                const float* conics, const float* opacities, int opac_per_camera, int32_t* cum_tiles,
                                int32_t* keep_scan, int32_t* tile_offsets, int32_t* tile_order, int64_t capacity_listed,
                                int64_t* stats, void* scratch, bool scratch_zeroed, int64_t* stats_mirror,
                                int64_t stats_seq, const MobgsTuning* tuning, void* stream) {
    const long long ng = (long long)C * N;
    const long long nt = (long long)C * tile_w * tile_h;
    const int heavy_len = tuning_heavy_len(tuning, (int)(nt < (1ll << 30) ? nt : (1ll << 30)));
    const int dense_hint = tuning_list_hint(tuning);
    if (C <= 0 || N < 0 || capacity < 1 || ng >= (1ll << 31) - 1 || nt >= (1ll << 31) - 1) {
        set_error("mobgs_isect_offsets: bad sizes C=%d N=%d tiles=%dx%d capacity=%d", C, N, tile_w, tile_h, capacity);
        return MOBGS_E_INVALID;
    }
    if (((uintptr_t)scratch & 7) != 0) {
        set_error("mobgs_isect_offsets: scratch must be 8-byte aligned");
        return MOBGS_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const int n = (int)ng;
    const IsectScratch L(scratch, (size_t)n, (size_t)nt, (size_t)capacity);
    if (!scratch_zeroed || n == 0)
        hipMemsetAsync(L.tile_count, 0, sizeof(int32_t) * L.zeroed_ints, st);  // tile counters, tickets, status words
    if (n == 0) {
        hipMemsetAsync(cum_tiles, 0, sizeof(int32_t), st);
        hipMemsetAsync(keep_scan, 0, 2 * sizeof(int32_t), st);  // base and first local of chunk 0
        hipMemsetAsync(stats, 0, 3 * sizeof(int64_t), st);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(tile_order ? 3 : 2), dim3(TSCAN_THREADS), 0, st, (int)nt, L.tile_count, L.tile_base, tile_offsets,
                           stats, tile_order, (int64_t)capacity, (int64_t)0, (int32_t*)nullptr, 0, heavy_len,
                           stats_mirror, stats_seq);
        return check_launch("isect_offsets(empty)");
    }
    // bounding-box counts -> cum_tiles; stats[0] = I_box
    hipLaunchKernelGGL(scan_lookback_kernel, dim3(L.nb1), dim3(SCAN_THREADS), 0, st, n, tiles_per_gauss, cum_tiles,
                       L.tickets, L.status1, stats, L.chunk_owner, L.owner_slots, (const int32_t*)nullptr, (int32_t*)nullptr);
    // keep flags, per-tile ranks and keep_scan over the first min(I_box, capacity) intersections (the caller
    // re-runs with a larger buffer when stats[0] > capacity); stats[1] = I_listed
    const int n_chunks = (capacity >> KEEP_CHUNK_LOG2) + 1;
    // LDS-ranked variant whenever one int per tile fits in LDS: measured faster at every grid size that qualifies
    // (scripts/ab/sweep_dense.sh: 576 tiles 47 -> 33 us, 1100 tiles 48 -> 40, 2040 tiles 50 -> 46, 5440 tiles 66.6 ->
    // 65.3), several times faster on dense image regions (long lists); larger grids keep the direct atomics
    (void)dense_hint;
    if (nt <= DENSE_MAX_TILES)
        hipLaunchKernelGGL((bin_kernel<true, false>), dim3(n_chunks), dim3(SCAN_THREADS), sizeof(int32_t) * (size_t)nt, st, n, N,
                           tile_w, tile_h, width, height, cull, capacity, cum_tiles, means2d, radii, conics, opacities,
                           opac_per_camera, L.chunk_cnt, L.owner, L.tile_of_j, L.rank_of_j, L.tile_count, keep_scan,
                           (int)nt, L.chunk_owner, (const float*)nullptr, (uint64_t*)nullptr, 0, 0, (const int32_t*)nullptr, (int)nt);
    else
        hipLaunchKernelGGL((bin_kernel<false, false>), dim3(n_chunks), dim3(SCAN_THREADS), 0, st, n, N, tile_w, tile_h, width,
                           height, cull, capacity, cum_tiles, means2d, radii, conics, opacities, opac_per_camera,
                           L.chunk_cnt, L.owner, L.tile_of_j, L.rank_of_j, L.tile_count, keep_scan, (int)nt,
                           L.chunk_owner, (const float*)nullptr, (uint64_t*)nullptr, 0, 0, (const int32_t*)nullptr, 0);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(tile_order ? 3 : 2), dim3(TSCAN_THREADS), 0, st, (int)nt, L.tile_count, L.tile_base, tile_offsets,
                       stats, tile_order, (int64_t)capacity, capacity_listed, keep_scan, n_chunks, heavy_len,
                       stats_mirror, stats_seq);
    return check_launch("isect_offsets");
}

extern "C" {

// shared by the synchronous entry point (stats_dev = NULL: the caller has read the counts) and the speculative one
static int emit_sort(int C, int N, int tile_w, int tile_h, int capacity, int64_t n_isects, int64_t max_tile_len,
                     const float* depths, const int32_t* cum_tiles, const int32_t* tile_offsets,
                     const void* offsets_scratch, uint64_t* sort_keys, int32_t* flatten_ids, uint64_t* isect_ids,
                     const int64_t* stats_dev, int64_t capacity_listed, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int tiles_per_cam = tile_w * tile_h;
    const int nt = C * tiles_per_cam;
    if (n_isects < 0 || n_isects >= (1ll << 31) - 1 || max_tile_len < 0) {
        set_error("mobgs_isect_emit_sort: n_isects=%lld out of range", (long long)n_isects);
        return MOBGS_E_INVALID;
    }
    if (n_isects == 0) return MOBGS_OK;
    const int n = C * N;
    // the compacted (owner, tile, rank) arrays pass A left in the scratch buffer of mobgs_isect_offsets
    const IsectScratch L(const_cast<void*>(offsets_scratch), (size_t)n, (size_t)nt, (size_t)capacity);
    hipLaunchKernelGGL(emit_kernel, dim3(4096), dim3(1024), 0, st, cum_tiles + n, L.chunk_cnt, L.owner, L.tile_of_j,
                       L.rank_of_j, depths, L.tile_base, nt, sort_keys, capacity, stats_dev, capacity_listed);
    // gsplat: tile_n_bits = floor(log2(n_tiles)) + 1
    int tile_bits = 0;
    while ((1ll << tile_bits) <= (long long)tiles_per_cam) ++tile_bits;
    // lists <= 2048 (all of them unless longer ones are expected): ONE launch, four tiles per workgroup, a wave per
    // list of <= 512 entries (registers), the workgroup for the few longer ones (16 KiB LDS).  Longer lists, when the
    // previous frame had any: 1024 threads, 128 KiB of the 160 KiB LDS (<= 16384 keys; beyond that in place in global
    // memory), in a separate launch over a compacted list of those tiles, so that the short lists keep their
    // occupancy.  The per-tile counters of pass A (dead since tile_scan) hold that list, a zeroed ticket word its length.
    const int small_cap = SHORT_SORT_LDS_KEYS, big_cap = 16384;
    const bool split = max_tile_len > small_cap;
    const int nmax_small = split ? small_cap : 0x7fffffff;
    // (max_tile_len: the previous frame's longest list, or this frame's in the synchronous form)
    auto sort_short = max_tile_len > 1024 ? tile_sort_short_kernel<32>
                      : max_tile_len > 512 ? tile_sort_short_kernel<16> : tile_sort_short_kernel<8>;
    hipLaunchKernelGGL(sort_short, dim3((nt + 3) / 4), dim3(256), 0, st, nt, tile_bits, tile_offsets, sort_keys,
                       flatten_ids, isect_ids, tiles_per_cam, nmax_small);
    if (split) {
        int32_t* long_ids = L.tile_count;
        int32_t* long_count = L.tickets + 1;
        hipLaunchKernelGGL(long_lists_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, nt, tile_offsets, small_cap,
                           long_ids, long_count);
        static_assert(sizeof(RadixShared) <= 160 * 1024, "radix sort state must fit the LDS of a CU");
        // Lists beyond the LDS radix sort (> 16384 entries): chunks + merge passes (see huge_chunk_sort_kernel) when the
        // previous frame's longest list says they are near -- and the dead (owner, tile, rank) triples of pass A, 12
        // bytes per bounding-box intersection, can hold a second copy of the keys; otherwise such a list takes the old
        // one-workgroup network in global memory (correct, slow; the next frame's hint then selects this path).
        const int64_t listed_cap = stats_dev ? capacity_listed : n_isects;
        const bool huge = max_tile_len > (3 * (int64_t)big_cap) / 4 && 8 * listed_cap <= 12 * (int64_t)capacity;
        hipLaunchKernelGGL(tile_sort_kernel<1024>, dim3(nt < 256 ? nt : 256), dim3(1024), sizeof(RadixShared), st,
                           nt, big_cap, tile_bits, tile_offsets, sort_keys, flatten_ids, isect_ids, tiles_per_cam,
                           small_cap + 1, huge ? big_cap : 0x7fffffff, long_ids, long_count);
        if (huge) {
            int32_t* max_chunks = L.tickets + 2;  // zeroed with the frame's counters
            uint64_t* tmp = reinterpret_cast<uint64_t*>(L.owner);
            hipLaunchKernelGGL(huge_chunk_sort_kernel, dim3(256), dim3(1024), sizeof(RadixShared), st, tile_offsets,
                               sort_keys, long_ids, long_count, max_chunks);
            int passes = 0;  // enough for one list holding every listed intersection
            while (((int64_t)big_cap << passes) < listed_cap) ++passes;
            for (int p = 0; p < passes; ++p)
                hipLaunchKernelGGL(huge_merge_kernel, dim3(256), dim3(256), 0, st, p, tile_offsets, sort_keys, tmp,
                                   long_ids, long_count, max_chunks);
            hipLaunchKernelGGL(huge_finish_kernel, dim3(256), dim3(256), 0, st, tile_bits, tiles_per_cam, tile_offsets,
                               sort_keys, tmp, flatten_ids, isect_ids, long_ids, long_count, max_chunks);
        }
    }
    return check_launch("isect_emit_sort");
}

int mobgs_isect_emit_sort(int C, int N, int tile_w, int tile_h, int capacity, int64_t n_isects,
                          int64_t max_tile_len, const float* depths, const int32_t* cum_tiles,
                          const int32_t* tile_offsets, const void* offsets_scratch, uint64_t* sort_keys,
                          int32_t* flatten_ids, uint64_t* isect_ids, void* stream) {
    return emit_sort(C, N, tile_w, tile_h, capacity, n_isects, max_tile_len, depths, cum_tiles, tile_offsets,
                     offsets_scratch, sort_keys, flatten_ids, isect_ids, nullptr, 0, stream);
}

int mobgs_isect_emit_sort_speculative(int C, int N, int tile_w, int tile_h, int capacity, int64_t capacity_listed,
                                      int64_t max_tile_len_hint, const float* depths, const int32_t* cum_tiles,
                                      const int32_t* tile_offsets, const int64_t* stats_dev,
                                      const void* offsets_scratch, uint64_t* sort_keys, int32_t* flatten_ids,
                                      uint64_t* isect_ids, void* stream) {
    if (!stats_dev || capacity_listed < 1) {
        set_error("mobgs_isect_emit_sort_speculative: stats_dev and capacity_listed are required");
        return MOBGS_E_INVALID;
    }
    return emit_sort(C, N, tile_w, tile_h, capacity, /*n_isects (unknown, > 0)*/ 1, max_tile_len_hint, depths,
                     cum_tiles, tile_offsets, offsets_scratch, sort_keys, flatten_ids, isect_ids, stats_dev,
                     capacity_listed, stream);
}

}  // extern "C"

// ---- fused single-pass lists (round 5) --------------------------------------------------------------------------------
extern "C" {

size_t mobgs_fused_seg_keys_len(int n_tiles, int seg_stride) {
    return (size_t)(n_tiles > 0 ? n_tiles : 0) * TC_COPIES * (size_t)(seg_stride > 0 ? seg_stride : 0);
}
int mobgs_fused_max_seg_stride(void) { return SHORT_SORT_LDS_KEYS; }

}  // extern "C"

float* mobgs::isect_bin_records(void* scratch, size_t n_gauss, size_t n_tiles, size_t capacity) {
    // the (owner, tile, rank) triples of the two-pass path are not written by the fused one: 3 * capacity ints, and
    // capacity >= 4 * n_gauss + 2 is checked by the launcher (12 floats per splat + the alignment slack)
    const IsectScratch L(scratch, n_gauss, n_tiles, capacity);
    return reinterpret_cast<float*>(((uintptr_t)L.owner + 15) & ~(uintptr_t)15);  // rows are read as float4
}

// scan -> bin (keys straight into the strided segments) -> offsets / schedule / counts -> per-tile sort: four launches
// (the two-pass path: five, with a second pass over the kept intersections and six dependent counter sweeps)
int mobgs::isect_fused_launch(int C, int N, int tile_w, int tile_h, int width, int height, int capacity,
                              const int32_t* tiles_per_gauss, int32_t* cum_tiles, int32_t* keep_scan,
                              int32_t* tile_offsets, int32_t* tile_order, int64_t capacity_listed, int64_t* stats,
                              void* scratch, int64_t* stats_mirror, int64_t stats_seq, uint64_t* seg_keys,
                              int seg_stride, int32_t* flatten_ids, uint64_t* isect_ids, int64_t max_tile_len_hint,
                              const int32_t* enum_order, const MobgsTuning* tuning, void* stream) {
    const long long ng = (long long)C * N;
    const long long nt = (long long)C * tile_w * tile_h;
    if (C <= 0 || N <= 0 || capacity < 1 || ng >= (1ll << 31) - 1 || nt >= (1ll << 31) - 1 || !seg_keys ||
        seg_stride < 1 || seg_stride > SHORT_SORT_LDS_KEYS || (long long)capacity < 4 * ng + 2 || capacity_listed < 1 ||
        tile_w > 0xFFFF || tile_h > 0xFFFF || C > 0xFFFF || ((uintptr_t)scratch & 127) != 0) {
        set_error("mobgs_project_and_bin_fused: bad sizes C=%d N=%d tiles=%dx%d capacity=%d seg_stride=%d", C, N, tile_w,
                  tile_h, capacity, seg_stride);
        return MOBGS_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const int n = (int)ng;
    const int heavy_len = tuning_heavy_len(tuning, (int)(nt < (1ll << 30) ? nt : (1ll << 30)));
    const IsectScratch L(scratch, (size_t)n, (size_t)nt, (size_t)capacity);
    const float* binrec = isect_bin_records(scratch, (size_t)n, (size_t)nt, (size_t)capacity);
    const int cstride = (int)count_stride((size_t)nt);
    // enum_order: the scan runs in the caller's splat order; its enumeration-order copy lives in the scratch buffer
    const int32_t* cum_search = enum_order ? L.cum_enum : cum_tiles;
    hipLaunchKernelGGL(scan_lookback_kernel, dim3(L.nb1), dim3(SCAN_THREADS), 0, st, n, tiles_per_gauss, cum_tiles,
                       L.tickets, L.status1, stats, L.chunk_owner, L.owner_slots, enum_order, L.cum_enum);
    const int n_chunks = (capacity >> KEEP_CHUNK_LOG2) + 1;
    // LDS-ranked variant: small grids (every workgroup touches most tiles several times) and scenes with long lists
    // (dense image regions: thousands of atomics on a few counters); on a large grid with short lists the plain
    // returning atomics are ahead (47.4 against 50.2 us at 5440 tiles / 300 k splats)
    // ... and with a (spatially coherent) enumeration order, whose whole point is that a workgroup's intersections
    // concentrate on few tiles
    // ... or with splats STORED in such an order (MobgsTuning.coherent_order: the caller's statement)
    // ... for a batch of cameras the table covers ONE camera's tiles (bin_kernel, dense_window)
    const long long dense_window = nt < (long long)tile_w * tile_h ? nt : (long long)tile_w * tile_h;
    if (dense_window <= DENSE_MAX_TILES &&
        (nt <= 2048 || max_tile_len_hint >= 1024 || enum_order || tuning_coherent_order(tuning)))
        hipLaunchKernelGGL((bin_kernel<true, true>), dim3(n_chunks), dim3(SCAN_THREADS), sizeof(int32_t) * (size_t)dense_window, st, n,
                           N, tile_w, tile_h, width, height, 1, capacity, cum_search, (const float*)nullptr,
                           (const int32_t*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, L.chunk_cnt, L.owner,
                           L.tile_of_j, L.rank_of_j, L.tile_count, keep_scan, (int)nt, L.chunk_owner, binrec, seg_keys,
                           seg_stride, cstride, enum_order, (int)dense_window);
    else
        hipLaunchKernelGGL((bin_kernel<false, true>), dim3(n_chunks), dim3(SCAN_THREADS), 0, st, n, N, tile_w, tile_h,
                           width, height, 1, capacity, cum_search, (const float*)nullptr, (const int32_t*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, 0, L.chunk_cnt, L.owner, L.tile_of_j,
                           L.rank_of_j, L.tile_count, keep_scan, (int)nt, L.chunk_owner, binrec, seg_keys, seg_stride,
                           cstride, enum_order, 0);
    hipLaunchKernelGGL(tile_finish_kernel, dim3(tile_order ? 3 : 2), dim3(TSCAN_THREADS), 0, st, (int)nt, L.tile_count,
                       cstride, tile_offsets, stats, tile_order, (int64_t)capacity, capacity_listed, seg_stride, keep_scan,
                       n_chunks, heavy_len, stats_mirror, stats_seq);
    const int tiles_per_cam = tile_w * tile_h;
    int tile_bits = 0;
    while ((1ll << tile_bits) <= (long long)tiles_per_cam) ++tile_bits;
    const int64_t longest = max_tile_len_hint > seg_stride ? seg_stride : max_tile_len_hint;
    auto sort_seg = longest > 1024 ? tile_sort_seg_kernel<32> : longest > 512 ? tile_sort_seg_kernel<16> : tile_sort_seg_kernel<8>;
    hipLaunchKernelGGL(sort_seg, dim3(((int)nt + 3) / 4), dim3(256), 0, st, (int)nt, tile_bits, tile_offsets, L.tile_count,
                       cstride, seg_keys, seg_stride, flatten_ids, isect_ids, tiles_per_cam);
    return check_launch("isect_fused");
}
