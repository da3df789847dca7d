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
// exclusive scan of int32 (3 launches: block sums, scan of sums, local scan + add)
// ---------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;  // 2048 ints per workgroup

__device__ inline int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(v, off, 64);
        if (lane >= off) v += n;
    }
    return v;
}

// inclusive scan across the 256-thread workgroup; returns this thread's inclusive value, *total = block sum
__device__ inline int block_incl_scan(int v, int* total) {
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int inc = wave_incl_scan(v, lane);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) base += (k < wv) ? wsum[k] : 0;
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return inc + base;
}

// n is read from device memory (*n_ptr, clamped to n_cap) so that a scan can follow a kernel that produced
// its own length without a host round trip; grids are sized for n_cap.
__device__ inline int scan_len(const int32_t* n_ptr, int n_cap) {
    const int n = n_ptr ? *n_ptr : n_cap;
    return n < n_cap ? n : n_cap;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_block_sums_kernel(const int32_t* __restrict__ n_ptr, int n_cap,
                                                                         const int32_t* __restrict__ in,
                                                                         int32_t* __restrict__ block_sums) {
    const int n = scan_len(n_ptr, n_cap);
    const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) s += (base + k < n) ? in[base + k] : 0;
    int total;
    block_incl_scan(s, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single workgroup: in-place exclusive scan of block_sums[nb]; the grand total goes to out[n] and stats_slot
__global__ void __launch_bounds__(SCAN_THREADS) scan_sums_kernel(const int32_t* __restrict__ n_ptr, int n_cap, int nb,
                                                                   int32_t* __restrict__ block_sums,
                                                                   int32_t* __restrict__ out,
                                                                   int64_t* __restrict__ stats_slot) {
    int carry = 0;
    for (int start = 0; start < nb; start += SCAN_THREADS) {
        const int i = start + threadIdx.x;
        const int v = (i < nb) ? block_sums[i] : 0;
        int total;
        const int inc = block_incl_scan(v, &total);
        if (i < nb) block_sums[i] = carry + inc - v;
        carry += total;
    }
    if (threadIdx.x == 0) {
        out[scan_len(n_ptr, n_cap)] = carry;
        if (stats_slot) *stats_slot = (int64_t)carry;
    }
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_apply_kernel(const int32_t* __restrict__ n_ptr, int n_cap,
                                                                    const int32_t* __restrict__ in,
                                                                    const int32_t* __restrict__ block_sums,
                                                                    int32_t* __restrict__ out) {
    const int n = scan_len(n_ptr, n_cap);
    const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    if (blockIdx.x * SCAN_BLOCK >= n) return;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    int total;
    const int inc = block_incl_scan(s, &total);
    int run = block_sums[blockIdx.x] + inc - s;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

// ---------------------------------------------------------------------------------------------------
// one thread per intersection: owner lookup (binary search in cum_tiles) + tile id
// ---------------------------------------------------------------------------------------------------
__device__ inline int owner_of(const int32_t* __restrict__ cum, int n, int j) {
    // largest g in [0,n) with cum[g] <= j   (cum is non-decreasing, cum[n] = I > j)
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] <= j)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__device__ inline int tile_of(int j, int g, int N, int tile_w, int tile_h, const int32_t* __restrict__ cum,
                              const float* __restrict__ means2d, const int32_t* __restrict__ radii) {
    const float2 m = reinterpret_cast<const float2*>(means2d)[g];
    const TileRect tr = tile_rect(m.x, m.y, radii[g], tile_w, tile_h);
    const int k = j - cum[g];
    const int w = tr.x1 - tr.x0;
    const int ty = tr.y0 + k / w, tx = tr.x0 + k % w;
    const int cam = g / N;
    return (cam * tile_h + ty) * tile_w + tx;
}

// Pass A, one thread per bounding-box intersection j: owner splat (binary search), tile, reach test.  A kept
// intersection takes its rank inside the tile's list with ONE returning atomic; (owner, tile, rank) are written
// out so that pass B is a pure streaming scatter (no second search, no second atomic).
__global__ void __launch_bounds__(256)
bin_kernel(int n_gauss, int N, int tile_w, int tile_h, int width, int height, int cull, int capacity,
           const int32_t* __restrict__ cum, const float* __restrict__ means2d, const int32_t* __restrict__ radii,
           const float* __restrict__ conics, const float* __restrict__ opacities, int opac_per_camera,
           int32_t* __restrict__ flags, int32_t* __restrict__ owner, int32_t* __restrict__ tile_of_j,
           int32_t* __restrict__ rank_of_j, int32_t* __restrict__ tile_count) {
    const int I = min(cum[n_gauss], capacity);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < I; j += gridDim.x * blockDim.x) {
        const int g = owner_of(cum, n_gauss, j);
        const int t = tile_of(j, g, N, tile_w, tile_h, cum, means2d, radii);
        int keep = 1;
        if (cull) {
            const float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
            const float op = opacities[opac_per_camera ? g : g % N];
            if (!(op * 255.f >= 1.f)) {
                keep = 0;  // alpha = min(0.999, op * exp(-sigma)) < 1/255 everywhere (sigma >= 0 where blended)
            } else if (ca > 0.f && cc > 0.f) {
                const int tl = t % (tile_w * tile_h);
                const int ty = tl / tile_w, tx = tl - ty * tile_w;
                const float x0 = (float)(tx * MOBGS_TILE) + 0.5f, y0 = (float)(ty * MOBGS_TILE) + 0.5f;
                const float x1 = fminf((float)(tx * MOBGS_TILE) + 15.5f, (float)width - 0.5f);
                const float y1 = fminf((float)(ty * MOBGS_TILE) + 15.5f, (float)height - 0.5f);
                const float2 m = reinterpret_cast<const float2*>(means2d)[g];
                const float smin = min_sigma_over_tile(m.x, m.y, ca, cb, cc, x0, x1, y0, y1);
                keep = (smin <= reach_threshold(op)) ? 1 : 0;
            }
        }
        flags[j] = keep;
        if (keep) {
            owner[j] = g;
            tile_of_j[j] = t;
            rank_of_j[j] = atomicAdd(&tile_count[t], 1);
        }
    }
}

// single workgroup: exclusive scan of tile_count[nt] -> tile_offsets[nt+1]; stats[2] = max count;
// tile_order[nt] (optional) = the tiles by DESCENDING list length (counting sort on ORDER_BUCKETS length classes):
// the order in which the compositing kernels hand tiles to workgroups, so that the longest lists start first and
// the waves of one workgroup get lists of similar length (longest-processing-time-first scheduling).  Only a
// schedule: any permutation gives the same images and gradients.
constexpr int ORDER_BUCKETS = 1024;
__global__ void __launch_bounds__(SCAN_THREADS) tile_scan_kernel(int nt, const int32_t* __restrict__ tile_count,
                                                                   int32_t* __restrict__ tile_offsets,
                                                                   int64_t* __restrict__ stats,
                                                                   int32_t* __restrict__ tile_order,
                                                                   int64_t capacity_box, int64_t capacity_listed) {
    __shared__ int smax[SCAN_THREADS];
    __shared__ int hist[ORDER_BUCKETS];
    int carry = 0, mx = 0;
    for (int start = 0; start < nt; start += SCAN_THREADS) {
        const int i = start + threadIdx.x;
        const int v = (i < nt) ? tile_count[i] : 0;
        mx = max(mx, v);
        int total;
        const int inc = block_incl_scan(v, &total);
        if (i < nt) tile_offsets[i] = carry + inc - v;
        carry += total;
    }
    smax[threadIdx.x] = mx;
    __syncthreads();
    for (int s = SCAN_THREADS / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) smax[threadIdx.x] = max(smax[threadIdx.x], smax[threadIdx.x + s]);
        __syncthreads();
    }
    const int longest = smax[0];
    if (threadIdx.x == 0) {
        tile_offsets[nt] = carry;
        stats[2] = (int64_t)longest;
    }
    // Arena too small (only checked when the caller runs ahead of the read-back, capacity_listed > 0): hand every
    // consumer EMPTY lists, so that kernels already enqueued behind this one touch nothing; stats keep the true
    // counts and the host redoes the binning with a larger arena.
    if (capacity_listed > 0 && (stats[0] > capacity_box || (int64_t)carry > capacity_listed)) {
        __syncthreads();
        for (int i = threadIdx.x; i <= nt; i += SCAN_THREADS) tile_offsets[i] = 0;
    }
    if (!tile_order) return;
    auto bucket = [&](int len) {
        const int q = longest > 0 ? (int)(((int64_t)len * (ORDER_BUCKETS - 1)) / longest) : 0;
        return ORDER_BUCKETS - 1 - q;  // bucket 0 = the longest lists
    };
    for (int b = threadIdx.x; b < ORDER_BUCKETS; b += SCAN_THREADS) hist[b] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += SCAN_THREADS) atomicAdd(&hist[bucket(tile_count[i])], 1);
    __syncthreads();
    {  // exclusive scan of hist: each thread owns ORDER_BUCKETS / SCAN_THREADS consecutive buckets
        constexpr int PER = ORDER_BUCKETS / SCAN_THREADS;
        int loc[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            loc[k] = hist[threadIdx.x * PER + k];
            sum += loc[k];
        }
        int total;
        int base = block_incl_scan(sum, &total) - sum;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            hist[threadIdx.x * PER + k] = base;
            base += loc[k];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += SCAN_THREADS) {
        const int pos = atomicAdd(&hist[bucket(tile_count[i])], 1);
        tile_order[pos] = i;
    }
}

// Pass B: streaming scatter of the 64-bit sort keys into the tile-contiguous segments
__global__ void __launch_bounds__(256) emit_kernel(const int32_t* __restrict__ n_box_ptr,
                                                     const int32_t* __restrict__ flags,
                                                     const int32_t* __restrict__ owner,
                                                     const int32_t* __restrict__ tile_of_j,
                                                     const int32_t* __restrict__ rank_of_j,
                                                     const float* __restrict__ depths,
                                                     const int32_t* __restrict__ tile_offsets,
                                                     uint64_t* __restrict__ sort_keys, int capacity,
                                                     const int64_t* __restrict__ stats, int64_t capacity_listed) {
    // speculative launch (stats != NULL): nothing to do when the arena was too small (see tile_scan_kernel)
    if (stats && (stats[0] > (int64_t)capacity || stats[1] > capacity_listed)) return;
    const int I = min(*n_box_ptr, capacity);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < I; j += gridDim.x * blockDim.x) {
        if (!flags[j]) continue;
        const int g = owner[j];
        const uint32_t db = __float_as_uint(depths[g]);
        sort_keys[(size_t)tile_offsets[tile_of_j[j]] + rank_of_j[j]] = ((uint64_t)db << 32) | (uint32_t)g;
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

template <int THREADS>
__global__ void __launch_bounds__(THREADS) tile_sort_kernel(int n_tiles_total, int lds_cap, int tile_bits,
                                                              const int32_t* __restrict__ tile_offsets,
                                                              uint64_t* __restrict__ sort_keys,
                                                              int32_t* __restrict__ flatten_ids,
                                                              uint64_t* __restrict__ isect_ids, int tiles_per_cam) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_keys[];
    const int t = blockIdx.x;
    if (t >= n_tiles_total) return;
    const int s = tile_offsets[t], e = tile_offsets[t + 1];
    const int n = e - s;
    if (n <= 0) return;
    uint64_t* seg = sort_keys + s;
    const int cam = t / tiles_per_cam, tl = t - cam * tiles_per_cam;
    const uint64_t hi_bits = (((uint64_t)cam << tile_bits) | (uint64_t)tl) << 32;
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

}  // namespace mobgs

using namespace mobgs;

extern "C" {

size_t mobgs_isect_scratch_bytes(int n_gauss, int n_tiles, int capacity) {
    const size_t nb1 = (size_t)(n_gauss + SCAN_BLOCK - 1) / SCAN_BLOCK + 1;
    const size_t nb2 = (size_t)(capacity + SCAN_BLOCK - 1) / SCAN_BLOCK + 1;
    return sizeof(int32_t) * (nb1 + nb2 + (size_t)n_tiles + 4 * (size_t)capacity + 32);
}

int mobgs_isect_offsets(int C, int N, int tile_w, int tile_h, int width, int height, int cull, int capacity,
                        const int32_t* tiles_per_gauss, const float* means2d, const int32_t* radii,
                        const float* conics, const float* opacities, int opac_per_camera, int32_t* cum_tiles,
                        int32_t* keep_scan, int32_t* tile_offsets, int32_t* tile_order, int64_t capacity_listed,
                        int64_t* stats, void* scratch, void* stream) {
    const long long ng = (long long)C * N;
    const long long nt = (long long)C * tile_w * tile_h;
    if (C <= 0 || N < 0 || capacity < 1 || ng >= (1ll << 31) - 1 || nt >= (1ll << 31) - 1) {
        set_error("mobgs_isect_offsets: bad sizes C=%d N=%d tiles=%dx%d capacity=%d", C, N, tile_w, tile_h, capacity);
        return MOBGS_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const int n = (int)ng;
    const int nb1 = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    const int nb2 = (capacity + SCAN_BLOCK - 1) / SCAN_BLOCK;
    int32_t* block_sums1 = (int32_t*)scratch;
    int32_t* block_sums2 = block_sums1 + nb1 + 1;
    int32_t* tile_count = block_sums2 + nb2 + 1;
    int32_t* flags = tile_count + nt;
    hipMemsetAsync(tile_count, 0, sizeof(int32_t) * nt, st);
    if (n == 0) {
        hipMemsetAsync(cum_tiles, 0, sizeof(int32_t), st);
        hipMemsetAsync(keep_scan, 0, sizeof(int32_t), st);
        hipMemsetAsync(stats, 0, 3 * sizeof(int64_t), st);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, (int)nt, tile_count, tile_offsets,
                           stats, tile_order, (int64_t)capacity, (int64_t)0);
        return check_launch("isect_offsets(empty)");
    }
    // bounding-box counts -> cum_tiles; stats[0] = I_box
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb1), dim3(SCAN_THREADS), 0, st, nullptr, n, tiles_per_gauss,
                       block_sums1);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, nullptr, n, nb1, block_sums1, cum_tiles,
                       stats);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb1), dim3(SCAN_THREADS), 0, st, nullptr, n, tiles_per_gauss,
                       block_sums1, cum_tiles);
    // keep flags + per-tile histogram (only the first `capacity` intersections; the caller re-runs with a larger
    // buffer when stats[0] > capacity)
    int32_t* owner = flags + capacity;
    int32_t* tile_of_j = owner + capacity;
    int32_t* rank_of_j = tile_of_j + capacity;
    hipLaunchKernelGGL(bin_kernel, dim3(4096), dim3(256), 0, st, n, N, tile_w, tile_h, width, height, cull, capacity,
                       cum_tiles, means2d, radii, conics, opacities, opac_per_camera, flags, owner, tile_of_j,
                       rank_of_j, tile_count);
    // keep_scan = exclusive scan of the flags over [0, min(I_box, capacity)); stats[1] = I_kept
    const int32_t* n_ptr = cum_tiles + n;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb2), dim3(SCAN_THREADS), 0, st, n_ptr, capacity, flags,
                       block_sums2);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, n_ptr, capacity, nb2, block_sums2,
                       keep_scan, stats + 1);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb2), dim3(SCAN_THREADS), 0, st, n_ptr, capacity, flags, block_sums2,
                       keep_scan);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, (int)nt, tile_count, tile_offsets,
                       stats, tile_order, (int64_t)capacity, capacity_listed);
    return check_launch("isect_offsets");
}

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
    // the (flag, owner, tile, rank) arrays pass A left in the scratch buffer of mobgs_isect_offsets
    const int nb1 = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    const int nb2 = (capacity + SCAN_BLOCK - 1) / SCAN_BLOCK;
    const int32_t* flags = (const int32_t*)offsets_scratch + (nb1 + 1) + (nb2 + 1) + nt;
    const int32_t* owner = flags + capacity;
    const int32_t* tile_of_j = owner + capacity;
    const int32_t* rank_of_j = tile_of_j + capacity;
    hipLaunchKernelGGL(emit_kernel, dim3(4096), dim3(256), 0, st, cum_tiles + n, flags, owner, tile_of_j, rank_of_j,
                       depths, tile_offsets, sort_keys, capacity, stats_dev, capacity_listed);
    // gsplat: tile_n_bits = floor(log2(n_tiles)) + 1
    int tile_bits = 0;
    while ((1ll << tile_bits) <= (long long)tiles_per_cam) ++tile_bits;
    if (max_tile_len <= 4096) {
        const int cap = 4096;
        hipLaunchKernelGGL(tile_sort_kernel<256>, dim3(nt), dim3(256), cap * sizeof(uint64_t), st, nt, cap, tile_bits,
                           tile_offsets, sort_keys, flatten_ids, isect_ids, tiles_per_cam);
    } else {
        const int cap = 16384;  // 128 KiB of the 160 KiB LDS; longer lists fall back to global memory
        hipLaunchKernelGGL(tile_sort_kernel<1024>, dim3(nt), dim3(1024), cap * sizeof(uint64_t), st, nt, cap,
                           tile_bits, tile_offsets, sort_keys, flatten_ids, isect_ids, tiles_per_cam);
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
