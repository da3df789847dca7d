// Backward pass of the HexPlane feature gather (BASELINE config #3) for gfx950.
//
// Differentiates deform.hip's hexplane_fwd_kernel, i.e. /root/reference/scene/hexplane.py:75-108,165-187: 3 levels
// x 6 planes of bilinear grid_sample(align_corners=True, padding_mode='border'), product over the planes of a level.
// The reference back-propagates it with torch's grid_sampler_2d_backward: one float atomicAdd per (point, plane, tap,
// channel) = 72 x 32 x N atomics (230 M at 100 k points).  On MI355X the L2 retires about one float atomic per
// channel and clock (128 per clock chip-wide), so that formulation is bound at ~1 ms per 100 k points no matter how it
// is issued (measured: 1.07 ms; it was round 1's kernel).  LDS float atomics are no way out either: ds_add_f32 costs
// ~170 cycles per wave instruction on this part (scripts/ubench/lds_atomic.hip: 45x an integer LDS atomic).
// The planes are small next to the stream -- 274 k cells for 7.2 M (point, plane, tap) hits, thousands of hits per cell
// on the time planes (every point of a view shares its time stamp) -- so equal cells are brought together first:
//
//   count     per (point, plane): integer atomic on the counter of its base cell (274 k counters).
//   scan      exclusive prefix sums over the counters: start of every cell's entry list (three small launches).
//   pass 1    a half-wave per point (32 lanes = 32 channels), as in the forward kernel: re-samples the 6 planes of a
//             level, forms for every plane  row[c] = v_feat[c] * prod_{other planes} sample[c]  and writes it, with
//             a 16-byte header (global cell id, the two bilinear fractions), into its slot of the cell's list
//             (slot = list start + a returning integer atomic on the cell's cursor: the 18 planes of a point are
//             issued by 18 lanes at once); the point / time gradients are finished here.
//   pass 2    the sorted entry list is cut into slices of 64 entries, one half-wave per slice: it streams its rows
//             (one coalesced 128-byte row per entry) and accumulates the four tap sums of a RUN of equal cells in
//             registers; only at the end of a run do 4 x 32 float atomics leave for the plane gradient: about one
//             run per touched cell instead of one per point, 34 M atomics instead of 230 M at 100 k points.
//
// Like torch's (and round 1's) formulation the result is a floating-point sum in an order that is not fixed.
#include <atomic>

#include "common.h"
#include "hexplane.h"

namespace mobgs {

constexpr int SLICE = 64;  // entries per half-wave in pass 2

struct CellMap {
    int base[18];   // first global cell id of the plane
    int total;      // number of cells of all planes
    // Planes with a time axis are "hot": every point of a view shares its time stamp, so all N points fall into one
    // or two rows of them (1500+ hits per cell at 100 k points).  Same-address atomics serialise in the L2, so their
    // counters / cursors are kept per workgroup in LDS (integer LDS atomics cost 4 cycles) and only block totals go
    // to the global ones.  lbase[id] = offset of the plane inside that LDS table, -1 for the spatial planes.
    int lbase[18];
    int hot_total;  // cells of all hot planes (0: no LDS table, everything through global atomics)
};
constexpr int PB = 512;            // points per workgroup of the count / pass-1 kernels
constexpr int WG = 1024;           // threads of those workgroups
constexpr int HOT_MAX = 24 * 1024; // LDS table entries (96 KB)

__device__ inline int cell_of(const CellMap& cm, const PlaneSet& planes, int id, const float q[4]) {
    const int p = id % 6;
    const int a = kAxisA[p], b = kAxisB[p];
    const float qa = a == 0 ? q[0] : (a == 1 ? q[1] : q[2]);
    const float qb = b == 1 ? q[1] : (b == 2 ? q[2] : q[3]);
    float ix = (qa + 1.f) * 0.5f * (float)(planes.ra[id] - 1);
    float iy = (qb + 1.f) * 0.5f * (float)(planes.rb[id] - 1);
    ix = fminf(fmaxf(ix, 0.f), (float)(planes.ra[id] - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(planes.rb[id] - 1));
    return (int)floorf(iy) * planes.ra[id] + (int)floorf(ix);
}
// global cell id of entry i of the LDS table
__device__ inline int hot_global_cell(const CellMap& cm, int i) {
    int id = 0;
#pragma unroll
    for (int k = 0; k < 18; ++k)
        if (cm.lbase[k] >= 0 && i >= cm.lbase[k]) id = k;
    return cm.base[id] + (i - cm.lbase[id]);
}

struct EntryHeader {  // 16 bytes
    int cell;         // global cell id of the base cell (plane base + y0 * ra + x0)
    float wx, wy;     // bilinear fractions
    int plane;
};

// scratch layout
struct BwdScratch {
    int* count;        // [total + 1]   entries per cell                  (zeroed by the launcher)
    int* cursor;       // [total + 1]   pass-1 write cursors              (zeroed by the launcher)
    int* start;        // [total + 1]   exclusive prefix of count
    int* block_sums;   // [ceil(total / 1024) + 1]
    EntryHeader* hdr;  // [18 N]
    float* rows;       // [18 N][32]
    size_t ints;
    __host__ __device__ BwdScratch(void* p, size_t total_cells, size_t n_points) {
        int* q = (int*)p;
        const size_t t1 = (total_cells + 1 + 63) & ~(size_t)63;
        count = q;
        cursor = q + t1;
        start = q + 2 * t1;
        block_sums = q + 3 * t1;
        ints = 3 * t1 + ((total_cells / 1024 + 2 + 63) & ~(size_t)63);
        hdr = (EntryHeader*)(q + ints);
        rows = (float*)(hdr + 18 * n_points);
    }
    static size_t bytes(size_t total_cells, size_t n_points) {
        BwdScratch s(nullptr, total_cells, n_points);
        return sizeof(int) * s.ints + 18 * n_points * (sizeof(EntryHeader) + 32 * sizeof(float)) + 256;
    }
};

// ---- count: entries per cell -----------------------------------------------------------------------------------
__global__ void __launch_bounds__(WG)
hex_count_kernel(int N, const float* __restrict__ pts, const float* __restrict__ times, const float* __restrict__ aabb,
                 PlaneSet planes, CellMap cm, int* __restrict__ count) {
    extern __shared__ int hot[];
    for (int i = threadIdx.x; i < cm.hot_total; i += WG) hot[i] = 0;
    __syncthreads();
    const int n0 = blockIdx.x * PB;
    for (int t = threadIdx.x; t < PB * 18; t += WG) {  // one (point, plane) pair per step
        const int n = n0 + t / 18, id = t % 18;
        if (n >= N) break;
        float q[4], dq[3];
        normalized_query(pts, times, aabb, n, q, dq);
        const int cell = cell_of(cm, planes, id, q);
        if (cm.hot_total > 0 && cm.lbase[id] >= 0)
            atomicAdd(&hot[cm.lbase[id] + cell], 1);
        else
            atomicAdd(&count[cm.base[id] + cell], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cm.hot_total; i += WG) {
        const int v = hot[i];
        if (v) atomicAdd(&count[hot_global_cell(cm, i)], v);
    }
}

// ---- scan: start[i] = sum of count[0 .. i) ------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) hex_scan_blocks_kernel(int n, const int* __restrict__ count, int* __restrict__ block_sums) {
    __shared__ int red[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    int v = i < n ? count[i] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k];
        block_sums[blockIdx.x] = s;
    }
}
__global__ void __launch_bounds__(1024) hex_scan_top_kernel(int nb, int* __restrict__ block_sums) {
    // exclusive scan of the block sums in place (one workgroup; nb <= a few thousand)
    __shared__ int carry;
    __shared__ int buf[1024];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const int v = i < nb ? block_sums[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int a = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < nb) block_sums[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[nb] = carry;
}
__global__ void __launch_bounds__(1024) hex_scan_final_kernel(int n, const int* __restrict__ count,
                                                              const int* __restrict__ block_sums, int* __restrict__ start) {
    __shared__ int buf[1024];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int v = i < n ? count[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int a = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
        __syncthreads();
        buf[threadIdx.x] += a;
        __syncthreads();
    }
    if (i < n) start[i] = block_sums[blockIdx.x] + buf[threadIdx.x] - v;
    if (i == n - 1) start[n] = block_sums[blockIdx.x] + buf[threadIdx.x];
}

__device__ inline float half_wave_sum(float v) {
    // sum over the 32 lanes of a half-wave (the 32 channels of one point)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));
    v += __shfl_xor(v, 16, 64);
    return v;
}

// ---- pass 1: per-point rows into the region lists; point / time gradients ---------------------------------------
__global__ void __launch_bounds__(WG)
hex_pass1_kernel(int N, const float* __restrict__ pts, const float* __restrict__ times, const float* __restrict__ aabb,
                 PlaneSet planes, CellMap cm, const float* __restrict__ v_feat, const int* __restrict__ start,
                 int* __restrict__ cursor, EntryHeader* __restrict__ hdr, float* __restrict__ rows,
                 float* __restrict__ v_pts, float* __restrict__ v_times) {
    extern __shared__ int hot[];  // hot planes: count of this block's points per cell, then the next free slot
    const int c = threadIdx.x & 31;
    const int n0 = blockIdx.x * PB;
    if (cm.hot_total > 0) {
        for (int i = threadIdx.x; i < cm.hot_total; i += WG) hot[i] = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < PB * 18; t += WG) {
            const int n = n0 + t / 18, id = t % 18;
            if (n >= N) break;
            if (cm.lbase[id] < 0) continue;
            float q[4], dq[3];
            normalized_query(pts, times, aabb, n, q, dq);
            atomicAdd(&hot[cm.lbase[id] + cell_of(cm, planes, id, q)], 1);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < cm.hot_total; i += WG) {  // one returning global atomic per touched cell
            const int v = hot[i];
            if (v) {
                const int gc = hot_global_cell(cm, i);
                hot[i] = start[gc] + atomicAdd(&cursor[gc], v);
            }
        }
        __syncthreads();
    }
    for (int n = n0 + (threadIdx.x >> 5); n < min(n0 + PB, N); n += WG / 32) {
        float q[4], dq[3];
        normalized_query(pts, times, aabb, n, q, dq);
        // lane id (< 18) reserves the slot of plane id in the list of its base cell: one returning integer atomic
        // instruction for the 18 planes of the point
        int my_slot = 0, my_cell = 0;
        if (c < 18) {
            const int cell = cell_of(cm, planes, c, q);
            my_cell = cm.base[c] + cell;
            if (cm.hot_total > 0 && cm.lbase[c] >= 0)
                my_slot = atomicAdd(&hot[cm.lbase[c] + cell], 1);  // LDS: next free slot of this block's range
            else
                my_slot = start[my_cell] + atomicAdd(&cursor[my_cell], 1);
        }
        float vq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            float s[6], dsx[6], dsy[6];
            Tap taps[6];
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const int id = l * 6 + p;
                taps[p] = make_tap(q[kAxisA[p]], q[kAxisB[p]], planes.ra[id], planes.rb[id]);
                const Tap& t = taps[p];
                const float* g = planes.p[id] + c;
                const float v00 = g[t.o00], v01 = g[t.o01], v10 = g[t.o10], v11 = g[t.o11];
                s[p] = (v00 * (1.f - t.wx) + v01 * t.wx) * (1.f - t.wy) + (v10 * (1.f - t.wx) + v11 * t.wx) * t.wy;
                dsx[p] = ((v01 - v00) * (1.f - t.wy) + (v11 - v10) * t.wy) * t.gx;
                dsy[p] = ((v10 - v00) * (1.f - t.wx) + (v11 - v01) * t.wx) * t.gy;
            }
            const float v = v_feat[(size_t)n * 96 + l * 32 + c];
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const int id = l * 6 + p;
                float others = v;
#pragma unroll
                for (int r = 0; r < 6; ++r)
                    if (r != p) others *= s[r];
                const int slot = __shfl(my_slot, id, 32);
                const int cell_id = __shfl(my_cell, id, 32);  // (all lanes take part in the shuffle)
                rows[(size_t)slot * 32 + c] = others;
                if (c == 0) {
                    const Tap& t = taps[p];
                    EntryHeader h;
                    h.cell = cell_id;
                    h.wx = t.wx;
                    h.wy = t.wy;
                    h.plane = id;
                    hdr[slot] = h;
                }
                vq[kAxisA[p]] += others * dsx[p];
                vq[kAxisB[p]] += others * dsy[p];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) vq[k] = half_wave_sum(vq[k]);
        if (c == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) v_pts[3 * n + k] += vq[k] * dq[k];
            v_times[n] = vq[3];
        }
    }
}

// ---- pass 2: run-length accumulation over the sorted entries -----------------------------------------------------
__global__ void __launch_bounds__(256)
hex_pass2_kernel(int n_entries, PlaneSet planes, CellMap cm, PlaneGradSet gplanes, const EntryHeader* __restrict__ hdr,
                 const float* __restrict__ rows) {
    const int c = threadIdx.x & 31;
    const int half = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int e0 = half * SLICE;
    if (e0 >= n_entries) return;
    const int e1 = min(e0 + SLICE, n_entries);
    float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
    int cur_cell = -1, cur_plane = 0;

    auto flush = [&]() {
        if (cur_cell < 0) return;
        const int ra = planes.ra[cur_plane], rb = planes.rb[cur_plane];
        const int local = cur_cell - cm.base[cur_plane];
        const int y0 = local / ra, x0 = local - y0 * ra;
        const int x1 = min(x0 + 1, ra - 1), y1 = min(y0 + 1, rb - 1);  // border clamp of grid_sample
        float* g = gplanes.p[cur_plane] + c;
        atomicAdd(g + ((size_t)y0 * ra + x0) * 32, a00);
        atomicAdd(g + ((size_t)y0 * ra + x1) * 32, a01);
        atomicAdd(g + ((size_t)y1 * ra + x0) * 32, a10);
        atomicAdd(g + ((size_t)y1 * ra + x1) * 32, a11);
    };

    EntryHeader h = hdr[e0];
    float v = rows[(size_t)e0 * 32 + c];
    for (int e = e0; e < e1; ++e) {
        EntryHeader hn = h;
        float vn = 0.f;
        if (e + 1 < e1) {  // next entry's loads in flight under this entry's arithmetic
            hn = hdr[e + 1];
            vn = rows[(size_t)(e + 1) * 32 + c];
        }
        if (h.cell != cur_cell) {  // half-wave uniform
            flush();
            cur_cell = h.cell;
            cur_plane = h.plane;
            a00 = a01 = a10 = a11 = 0.f;
        }
        const float ax = 1.f - h.wx, ay = 1.f - h.wy;
        a00 = __fmaf_rn(v, ax * ay, a00);
        a01 = __fmaf_rn(v, h.wx * ay, a01);
        a10 = __fmaf_rn(v, ax * h.wy, a10);
        a11 = __fmaf_rn(v, h.wx * h.wy, a11);
        h = hn;
        v = vn;
    }
    flush();
}

static void allow_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
}

}  // namespace mobgs

using namespace mobgs;

extern "C" {

static size_t total_cells(const int32_t* ra_host, const int32_t* rb_host) {
    size_t t = 0;
    for (int i = 0; i < 18; ++i) t += (size_t)ra_host[i] * (size_t)rb_host[i];
    return t;
}

size_t mobgs_hexplane_bwd_scratch_bytes(int N, const int32_t* ra_host, const int32_t* rb_host) {
    if (!ra_host || !rb_host) return 0;
    return BwdScratch::bytes(total_cells(ra_host, rb_host), (size_t)(N < 0 ? 0 : N));
}

int mobgs_hexplane_bwd(int N, const float* pts, const float* times, const float* aabb,
                       const float* const* planes_host, const int32_t* ra_host, const int32_t* rb_host,
                       const float* v_feat, float* const* gplanes_host, float* v_pts, float* v_times, void* scratch,
                       void* stream) {
    PlaneSet ps;
    PlaneGradSet gs;
    CellMap cm;
    if (N < 0 || !planes_host || !ra_host || !rb_host || !gplanes_host) {
        set_error("mobgs_hexplane_bwd: bad arguments");
        return MOBGS_E_INVALID;
    }
    long long total = 0;
    for (int i = 0; i < 18; ++i) {
        ps.p[i] = planes_host[i];
        ps.ra[i] = ra_host[i];
        ps.rb[i] = rb_host[i];
        gs.p[i] = gplanes_host[i];
        if (!ps.p[i] || !gs.p[i] || ps.ra[i] < 1 || ps.rb[i] < 1) {
            set_error("mobgs_hexplane_bwd: bad plane %d", i);
            return MOBGS_E_INVALID;
        }
        cm.base[i] = (int)total;
        total += (long long)ps.ra[i] * ps.rb[i];
    }
    // planes (a, 3): p in {2, 4, 5} of every level
    long long hot = 0;
    for (int i = 0; i < 18; ++i) {
        const int p = i % 6;
        const bool has_time = p == 2 || p == 4 || p == 5;
        cm.lbase[i] = has_time ? (int)hot : -1;
        if (has_time) hot += (long long)ps.ra[i] * ps.rb[i];
    }
    if (hot > HOT_MAX) {  // time planes too large for the LDS table: global atomics for everything
        hot = 0;
        for (int i = 0; i < 18; ++i) cm.lbase[i] = -1;
    }
    cm.hot_total = (int)hot;
    if (total >= (1ll << 30) || 18ll * N >= (1ll << 31) - 1) {
        set_error("mobgs_hexplane_bwd: %lld plane cells / %d points exceed the 32-bit entry index", total, N);
        return MOBGS_E_UNSUPPORTED;
    }
    cm.total = (int)total;
    if (N == 0) return MOBGS_OK;
    if (!scratch || ((uintptr_t)scratch & 15) != 0) {
        set_error("mobgs_hexplane_bwd: scratch of mobgs_hexplane_bwd_scratch_bytes() bytes, 16-byte aligned, required");
        return MOBGS_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    BwdScratch S(scratch, (size_t)total, (size_t)N);
    (void)hipMemsetAsync(S.count, 0, sizeof(int) * (size_t)(S.start - S.count), st);  // count + cursor
    const int n_entries = 18 * N;
    const int n_blocks = (N + PB - 1) / PB;
    const size_t hot_bytes = sizeof(int) * (size_t)cm.hot_total;
    static std::atomic<unsigned long long> done_c{0}, done_p{0};
    allow_lds(reinterpret_cast<const void*>(hex_count_kernel), (int)sizeof(int) * HOT_MAX, done_c);
    allow_lds(reinterpret_cast<const void*>(hex_pass1_kernel), (int)sizeof(int) * HOT_MAX, done_p);
    hipLaunchKernelGGL(hex_count_kernel, dim3(n_blocks), dim3(WG), hot_bytes, st, N, pts, times, aabb, ps, cm, S.count);
    const int nb = (cm.total + 1023) / 1024;
    hipLaunchKernelGGL(hex_scan_blocks_kernel, dim3(nb), dim3(1024), 0, st, cm.total, S.count, S.block_sums);
    hipLaunchKernelGGL(hex_scan_top_kernel, dim3(1), dim3(1024), 0, st, nb, S.block_sums);
    hipLaunchKernelGGL(hex_scan_final_kernel, dim3(nb), dim3(1024), 0, st, cm.total, S.count, S.block_sums, S.start);
    hipLaunchKernelGGL(hex_pass1_kernel, dim3(n_blocks), dim3(WG), hot_bytes, st, N, pts, times, aabb, ps, cm, v_feat,
                       S.start, S.cursor, S.hdr, S.rows, v_pts, v_times);
    const int n_half = (n_entries + SLICE - 1) / SLICE;
    hipLaunchKernelGGL(hex_pass2_kernel, dim3((n_half + 7) / 8), dim3(256), 0, st, n_entries, ps, cm, gs, S.hdr, S.rows);
    return check_launch("hexplane_bwd");
}

}  // extern "C"
