// K7 (round 4): backward compositing with the per-splat gradient SUMS on the matrix pipe.
//
// Semantics: gsplat v1.4.0 rasterize_to_pixels_bwd [upstream, SURVEY.md Appendix A.4], reached from the reference at
// /root/reference/gaussian_renderer/__init__.py:201-217 (and the other rasterization() calls listed in raster.hip).
//
// raster_bwd_kernel (raster.hip) keeps 6 + CD gradient sums per lane and per list entry, adds 16 + 14 FMAs per
// (pixel, splat) pair into them and pays a 64-lane reduction (12 permlane swaps + 8 DPP adds + the clears) per entry:
// 176 issued lane-operations per live pair, of which the arithmetic of the pair is ~50.  Here the per-pixel arithmetic
// is unchanged (eval_splat, the reciprocal, T, `behind`, v_alpha, v_sigma: bit-identical per pixel) but a pair
// only PRODUCES its two weights
//       fac = alpha * T                (weight of the colour cotangent:  v_colour[e] += fac * v_out[p])
//       vs  = v_sigma                  (weight of the geometry terms:    v_xy, v_conic, v_opacity)
// and the sums over the pixels are matrix products with K = pixel:
//       G[16 x 16] += A[16 x 4] * B[4 x 16]     v_mfma_f32_16x16x4_f32 (exact fp32 FMAs), 16 of them per 64 pixels
//   rows    0..7  = fac of 8 list entries, rows 8..15 = vs of the same 8 entries
//   columns 0..9  = the pixel's colour cotangents v_out[p][c]          (used by the fac rows)
//   columns 10..15 = the pixel's moments {1, cx, cy, cx^2, cx cy, cy^2} (used by the vs rows; cx, cy = pixel centre
//                    relative to the TILE centre, |c| <= 7.5, so the moment sums stay small)
// so one accumulator block carries both products, B (16 registers per quadrant) is built once per tile and never
// changes, and the wave reduction, the sixteen per-lane accumulators and their clears are gone.  From the moments
//       S0 = sum vs, Sx = sum vs cx, ...   with m = splat centre relative to the tile centre, d = m - c:
//       v_x  = ca (m_x S0 - Sx) + cb (m_y S0 - Sy)          v_conic_a = (m_x (m_x S0 - 2 Sx) + Sxx) / 2
//       v_y  = cb (m_x S0 - Sx) + cc (m_y S0 - Sy)          v_conic_b = m_x m_y S0 - m_x Sy - m_y Sx + Sxy
//       v_opacity = -S0 / opacity   (vs = -opacity vis v_alpha on every pair where the 0.999 clamp is not active)
//
// Data flow of one wave (one 16x16 tile, 4 pixels per lane = one per 8x8 quadrant, as in raster.hip):
//   * 64 list entries per batch, lane = entry: id, reach byte, gradient slot and the geometric half of the record stay
//     in that lane's registers -- no LDS slab.  The record of the entry being evaluated is fetched with SCALAR loads
//     (s_load_dwordx4 through the constant cache, one batch entry ahead) and feeds the VALU as SGPR operands: the LDS
//     pipe, which the four broadcast reads per (entry, quadrant) would saturate here, only carries the weights;
//   * per quadrant k the entries that can reach it (ballot of the reach bits, cut at the quadrant's last blended
//     entry) are walked back to front; an evaluation with at least one passing pixel writes its two weight rows into the
//     wave's 16 x 64 LDS tile (XOR-swizzled 16-byte slots: conflict-free stores by pixel, conflict-free ds_read_b128 by
//     (row, pixel block)); after 8 such rows -- or at the end of the quadrant's walk -- the tile is read back in the A
//     layout (lane = (row, 16-pixel block)), 16 MFMAs form the 8 entries' sums over the quadrant, and the result is
//     added to a per-batch accumulator acc[64 entries][16] in LDS (plain read-add-write: one wave, LDS in order);
//   * after the four quadrants lane = entry reads its 16 sums, converts the moments and stores ONE 64-byte gradient
//     record to the entry's slot -- every listed entry of a walked batch gets a record (zeros if nothing blended it).
// Heavy tiles (4 waves per tile, one quadrant each): every wave keeps its own accumulator; the workgroup sums the four
// in fixed quadrant order after a barrier.  No floating-point atomics anywhere: gradients are bit-reproducible.
#include <type_traits>

#include "common.h"
#include "raster_shared.h"

namespace mobgs {

typedef float mf4 __attribute__((ext_vector_type(4)));

#ifndef MOBGS_MFMA_WAVES
#define MOBGS_MFMA_WAVES 3  // waves per SIMD the register allocation of the wave-per-tile kernel aims at
#endif
#ifndef MOBGS_TEAM_WAVES
#define MOBGS_TEAM_WAVES 4  // ... of the team kernel (33 KiB of LDS per workgroup: 4 workgroups per CU)
#endif

// development aid (-DMOBGS_MFMA_TIMING): shader cycles per phase, summed over all waves, in g_mfma_timing --
// [0] prologue, [1] staging, [2] quadrant walks without the group reductions, [3] group reductions, [4] flush,
// [5] waves, [6] groups, [7] evaluations.  Read with mobgs_debug_mfma_timing (not part of the ABI).
#ifdef MOBGS_MFMA_TIMING
__device__ unsigned long long g_mfma_timing[8];
#define MFMA_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define MFMA_ACC(i, expr) t_acc[i] += (expr)
#else
#define MFMA_T(var)
#define MFMA_ACC(i, expr)
#endif

constexpr int GROUP = 8;  // list entries per MFMA group (rows 0..7 fac, 8..15 v_sigma)

// uniform (scalar) load of one packed splat record: `g` must be wave-uniform
#ifdef MOBGS_MFMA_FAKE_SMEM  // experiment: every evaluation fetches the batch's FIRST record (always a cache hit; wrong results)
#define MFMA_LANE(j) 0
#else
#define MFMA_LANE(j) (j)
#endif
template <int RS>
__device__ __forceinline__ void load_record_uniform(const float* records, int g, float (&rec)[RS]) {
    typedef __attribute__((address_space(4))) const mf4 cmf4;
    cmf4* p = (cmf4*)(unsigned long long)(records + (size_t)g * RS);
#pragma unroll
    for (int q = 0; q < RS / 4; ++q) {
        const mf4 v = p[q];
        rec[4 * q] = v[0];
        rec[4 * q + 1] = v[1];
        rec[4 * q + 2] = v[2];
        rec[4 * q + 3] = v[3];
    }
}

// ---- what one wave needs to turn pair weights into sums ----------------------------------------------------------------
// Weight tile: 16 rows of 64 floats (row r = 256 bytes; 4096-aligned so that OR / XOR compose addresses); the 16-byte
// slot sl of row r is stored at slot sl ^ pi(r), pi(r) = (r + 4) & 15: stores by pixel (lane = pixel, row uniform) and
// ds_read_b128 by (row = lane & 15, pixel block = lane >> 4) are both conflict-free.
// Accumulator: acc[batch position][STRIDE] floats, columns 0..5 = moment sums, 6..15 = colour sums.
template <int CD, int STRIDE>
struct MfmaWave {
    unsigned w_store;  // lane's store address in row 0 before the row swizzle
    unsigned a_load;   // lane's first A-operand read (^ (t << 4) for the other three)
    float* acc;        // this wave's accumulator block
    int* rowpos;       // [GROUP]: batch position of the entry in each row of the current group
    int d_half;        // which 4 rows of the 8 (fac) / 8 (v_sigma) this lane holds in the result block
    int d_col;         // accumulator column this lane adds
    bool d_lane;       // lane holds a used column
    int lane;

    __device__ __forceinline__ void init(float* w_tile, float* acc_block, int* rowpos_block, int lane_) {
        lane = lane_;
        const int bq = lane >> 4, bj = lane & 15;
        const unsigned w_base = (unsigned)(size_t)w_tile;
        w_store = w_base | (unsigned)(lane << 2);
        a_load = w_base | (unsigned)(bj << 8) | (unsigned)((((bq << 2) ^ ((bj + 4) & 15)) & 15) << 4);
        acc = acc_block;
        rowpos = rowpos_block;
        d_half = bq & 1;
        const bool colour = bq < 2;
        d_lane = colour ? bj < CD : bj >= 10;
        // lanes without a used column (bj >= CD in the colour rows, bj < 10 in the moment rows) still run reduce_group's
        // four accumulator loads: keep their column INSIDE the record (0) instead of 6 + bj up to 21 / bj - 10 down to -10,
        // which only stayed inside the shared struct by the accident of its layout
        d_col = d_lane ? (colour ? 6 + bj : bj - 10) : 0;
        static_assert(STRIDE >= 16, "an accumulator row holds the 16 components a used column can select");
        if (lane < GROUP) rowpos[lane] = 0;
    }

    // one group of `rows` <= 8 entries: transposed read, 16 MFMAs against the quadrant's B operands, add into the batch
    // accumulator.  The rows' batch positions and the accumulator words they select are fetched BEFORE the MFMAs are
    // issued, so those two LDS round trips run under the matrix work instead of behind it.
    __device__ __forceinline__ void reduce_group(const float (&Bk)[16], int rows) const {
        wave_lds_fence();
        mf4 A[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) A[t] = *(const __attribute__((address_space(3))) mf4*)(unsigned long long)(a_load ^ (unsigned)(t << 4));
        const int4 pos = *reinterpret_cast<const int4*>(rowpos + 4 * d_half);
        float* a0 = acc + pos.x * STRIDE + d_col;
        float* a1 = acc + pos.y * STRIDE + d_col;
        float* a2 = acc + pos.z * STRIDE + d_col;
        float* a3 = acc + pos.w * STRIDE + d_col;
        const float o0 = *a0, o1 = *a1, o2 = *a2, o3 = *a3;
        mf4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mm = 0; mm < 16; mm += 2) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mm >> 2][mm & 3], Bk[mm], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mm >> 2][(mm + 1) & 3], Bk[mm + 1], d1, 0, 0, 0);
        }
        const int row0 = 4 * d_half;
        if (d_lane) {  // rows of this lane's half that exist in the group
            if (row0 < rows) *a0 = o0 + (d0[0] + d1[0]);
            if (row0 + 1 < rows) *a1 = o1 + (d0[1] + d1[1]);
            if (row0 + 2 < rows) *a2 = o2 + (d0[2] + d1[2]);
            if (row0 + 3 < rows) *a3 = o3 + (d0[3] + d1[3]);
        }
        wave_lds_fence();
    }
};

// B operands of one quadrant: lane (bq = lane >> 4, bj = lane & 15) holds, for MFMA m, column bj of pixel 16 bq + m --
// bj < CD: the pixel's colour cotangent bj; bj >= 10: its moment bj - 10 (pixel centre relative to the tile centre)
template <int CD>
__device__ __forceinline__ void build_b_operands(float (&Bk)[16], int qd, int lane, int cam, int tx, int ty, int width,
                                                 int height, const float* __restrict__ v_render) {
    const int bq = lane >> 4, bj = lane & 15;
    const int col = bj < CD ? bj : 0;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int p = 16 * bq + m;
        const int lx = 8 * (qd & 1) + (p & 7), ly = 8 * (qd >> 1) + (p >> 3);
        const int pxi = tx * MOBGS_TILE + lx, pyi = ty * MOBGS_TILE + ly;
        const float cx = (float)lx - 7.5f, cy = (float)ly - 7.5f;
        // unconditional load from a clamped address (no branch per operand), then a select
        const int cxi = min(pxi, width - 1), cyi = min(pyi, height - 1);
        const float vr = v_render[(((size_t)cam * height + cyi) * width + cxi) * CD + col];
        const bool inside = pxi < width && pyi < height;
        const float mom = bj == 10 ? 1.f : bj == 11 ? cx : bj == 12 ? cy : bj == 13 ? cx * cx : bj == 14 ? cx * cy : cy * cy;
        Bk[m] = bj < CD ? (inside ? vr : 0.f) : (bj >= 10 ? mom : 0.f);
    }
}

// moments -> slot record of one (tile, entry) pair in the format of blend_bwd (raster.hip): the RAW geometric sums
// {sum v_sigma dx, sum v_sigma dy, sum v_sigma dx^2, sum v_sigma dx dy, sum v_sigma dy^2, sum v_sigma}, v_colour[..] --
// conic and opacity enter once per splat in stage 2 (finish_geometry)
__device__ __forceinline__ void convert_moments(const float (&a)[16], float gx, float gy, float /*ca*/, float /*cb*/,
                                                float /*cc*/, float op, float mcx, float mcy, float (&out)[16]) {
    const float S0 = a[0], Sx = a[1], Sy = a[2], Sxx = a[3], Sxy = a[4], Syy = a[5];
    const float mx = gx - mcx, my = gy - mcy;
    const float Dx = __fmaf_rn(mx, S0, -Sx), Dy = __fmaf_rn(my, S0, -Sy);
    out[0] = Dx;
    out[1] = Dy;
    out[2] = __fmaf_rn(mx, Dx - Sx, Sxx);
    out[3] = __fmaf_rn(mx, Dy, __fmaf_rn(-my, Sx, Sxy));
    out[4] = __fmaf_rn(my, Dy - Sy, Syy);
    out[5] = S0;
#pragma unroll
    for (int c = 6; c < 16; ++c) out[c] = a[c];
}

// Walk the entries of batch `hi` whose bit is set in m (bit j = batch position j = list index hi - j) for ONE quadrant,
// back to front.  g = this lane's staged flat id (lane = batch position).  Per-pixel state of the quadrant by reference.
template <int CD, int STRIDE, int RS>
__device__ __forceinline__ void walk_quadrant(const MfmaWave<CD, STRIDE>& mw, const float* __restrict__ records, int g,
                                              unsigned long long m, int hi, float px, float py, int binf, float& T,
                                              float& behind, float tvab, const float (&vo)[CD], const float (&Bk)[16]
#ifdef MOBGS_MFMA_TIMING
                                              , unsigned long long (&t_acc)[8]
#endif
) {
    int nrow = 0;
    // one (entry, quadrant) evaluation; rec = the entry's record in SGPRs, jc = its position in the batch
    auto evaluate = [&](const float (&rec)[RS], int jc) {
        MFMA_ACC(7, 1);
        const Eval ev = eval_splat(rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], px, py);
        const bool pass = ev.pass && (hi - jc <= binf);
        if (__builtin_amdgcn_ballot_w64(pass) == 0ull) return;
        // the pair arithmetic of blend_bwd (raster.hip), minus the sums
        const float alpha = pass ? ev.alpha : 0.f;
        const float om = 1.f - alpha;
        const float ra = __builtin_amdgcn_rcpf(om);   // (no Newton step: raster.hip, blend_bwd)
        T *= ra;
        const float fac = alpha * T;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < CD; ++c) dot = __fmaf_rn(rec[6 + c], vo[c], dot);
        const float v_alpha = __fmaf_rn(T, dot, ra * (tvab - behind));
        const float ov = ev.raw;
        const bool live = pass && ov <= ALPHA_MAX;
        const float v_sigma = live ? -ov * v_alpha : 0.f;
        behind = __fmaf_rn(fac, dot, behind);
        // rows nrow (fac) and 8 + nrow (v_sigma) of the weight tile, column = this lane's pixel
        const unsigned sw = (unsigned)((nrow << 8) | (((nrow + 4) & 15) << 4));
        const unsigned a_fac = mw.w_store ^ sw;
        *(__attribute__((address_space(3))) float*)(unsigned long long)(a_fac) = fac;
        *(__attribute__((address_space(3))) float*)(unsigned long long)(a_fac ^ 0x880u) = v_sigma;
        if (mw.lane == 0) mw.rowpos[nrow] = jc;
        if (++nrow == GROUP) {
            MFMA_T(t_r0);
            mw.reduce_group(Bk, GROUP);
            MFMA_T(t_r1);
            MFMA_ACC(3, t_r1 - t_r0);
            MFMA_ACC(6, 1);
            nrow = 0;
        }
    };
    // the record of the NEXT entry is fetched (scalar loads) while the current one is evaluated -- two register sets used
    // alternately, so nothing is copied.  lgkmcnt cannot tell SMEM loads apart (they return out of order), so the wait
    // for the CURRENT record must come before the NEXT one is issued: the empty asm makes the compiler place it there.
    float rec_a[RS], rec_b[RS];
    int ja = __builtin_ctzll(m), jb = 0;
    m &= m - 1ull;
    load_record_uniform<RS>(records, __builtin_amdgcn_readlane(g, MFMA_LANE(ja)), rec_a);
    while (true) {
        asm volatile("" ::"s"(rec_a[0]) : "memory");
        const bool more_b = m != 0ull;
        if (more_b) {
            jb = __builtin_ctzll(m);
            m &= m - 1ull;
            load_record_uniform<RS>(records, __builtin_amdgcn_readlane(g, MFMA_LANE(jb)), rec_b);
        }
        evaluate(rec_a, ja);
        if (!more_b) break;
        asm volatile("" ::"s"(rec_b[0]) : "memory");
        const bool more_a = m != 0ull;
        if (more_a) {
            ja = __builtin_ctzll(m);
            m &= m - 1ull;
            load_record_uniform<RS>(records, __builtin_amdgcn_readlane(g, MFMA_LANE(ja)), rec_a);
        }
        evaluate(rec_b, jb);
        if (!more_a) break;
    }
    if (nrow > 0) {
        MFMA_T(t_r0);
        mw.reduce_group(Bk, nrow);
        MFMA_T(t_r1);
        MFMA_ACC(3, t_r1 - t_r0);
        MFMA_ACC(6, 1);
    }
}
#ifdef MOBGS_MFMA_TIMING
#define MFMA_TARG , t_acc
#else
#define MFMA_TARG
#endif

// per-pixel state of one quadrant, lane = pixel (lane & 7, lane >> 3); loads from clamped addresses + selects
template <int CD>
struct QuadState {
    float px, py, T, behind, tvab;
    float vo[CD];
    int binf;
    __device__ __forceinline__ void load(int qd, int lane, int cam, int tx, int ty, int width, int height,
                                         const float* __restrict__ backgrounds, const float* __restrict__ render_alphas,
                                         const int32_t* __restrict__ last_ids, const float* __restrict__ v_render,
                                         const float* __restrict__ v_alphas) {
        const int pxi = tx * MOBGS_TILE + 8 * (qd & 1) + (lane & 7);
        const int pyi = ty * MOBGS_TILE + 8 * (qd >> 1) + (lane >> 3);
        px = (float)pxi + 0.5f;
        py = (float)pyi + 0.5f;
        const bool inside = pxi < width && pyi < height;
        const size_t pix = ((size_t)cam * height + min(pyi, height - 1)) * width + min(pxi, width - 1);
        binf = inside ? last_ids[pix] : -1;  // pixels outside the image never become valid
        const float Tf = inside ? 1.f - render_alphas[pix] : 1.f;
        const float va = (inside && v_alphas) ? v_alphas[pix] : 0.f;
        const float* vr = v_render + pix * CD;
        float bgdot = 0.f;
        bool nz = va != 0.f;
#pragma unroll
        for (int c = 0; c < CD; ++c) {
            vo[c] = inside ? vr[c] : 0.f;
            nz = nz || (vo[c] != 0.f);
        }
        if (backgrounds) {
#pragma unroll
            for (int c = 0; c < CD; ++c) bgdot = __fmaf_rn(backgrounds[cam * CD + c], vo[c], bgdot);
        }
        // a pixel whose cotangents are all exactly zero contributes to no gradient: treat it like one that blended
        // nothing, so that it neither extends the walk nor passes a test
        if (!nz) binf = -1;
        T = Tf;
        behind = 0.f;
        tvab = Tf * (va - bgdot);
    }
};

__device__ __forceinline__ int wave_max_i(int t) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t = max(t, __shfl_xor(t, off, 64));
    return __builtin_amdgcn_readfirstlane(t);
}

// =====================================================================================================================
// (1) one wave per tile, four pixels per lane
// =====================================================================================================================
constexpr int ACC_STRIDE = 20;  // floats per accumulator row: 16 sums + 4 pad (80-byte rows: conflict-free b128 by lane)

struct BwdMfmaShared {
    float w[TILES_PER_WG][16][64] __attribute__((aligned(4096)));  // weight tile of each wave (4 KiB, swizzled)
    float acc[TILES_PER_WG][64][ACC_STRIDE] __attribute__((aligned(16)));
    int rowpos[TILES_PER_WG][GROUP] __attribute__((aligned(16)));
};

template <int CD, bool FILTER>
__device__ __forceinline__ void composite_bwd_mfma(
    int tile, int wv, int lane, BwdMfmaShared& sh, ClassSel cls, int tile_w, int tile_h, int width, int height,
    const float* __restrict__ records, const float* __restrict__ backgrounds, const int32_t* __restrict__ radii,
    const int32_t* __restrict__ cum_tiles, const int32_t* __restrict__ keep_scan,
    const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
    const float* __restrict__ render_alphas, const int32_t* __restrict__ last_ids, const float* __restrict__ v_render,
    const float* __restrict__ v_alphas, float* __restrict__ grad_slots, const uint8_t* __restrict__ isect_reach,
    int32_t* __restrict__ any_record) {
    static_assert(CD >= 1 && CD <= 10, "one accumulator block: 10 colour columns + 6 moment columns");
    constexpr int RS = (6 + CD + 3) & ~3;
    constexpr int RQ = RS / 4;
    constexpr int NP = 4;
    const int tiles_per_cam = tile_w * tile_h;
    const int cam = tile / tiles_per_cam;
    const int tl = tile - cam * tiles_per_cam;
    const int ty = tl / tile_w, tx = tl - ty * tile_w;

    const int s = __builtin_amdgcn_readfirstlane(tile_offsets[tile]);
    const int e = __builtin_amdgcn_readfirstlane(tile_offsets[tile + 1]);
    if (e <= s) return;
#ifdef MOBGS_MFMA_TIMING
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 1, 0, 0};
#endif
    MFMA_T(t_begin);

    QuadState<CD> qs[NP];
    int topk[NP];
    int top = -1;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        qs[k].load(k, lane, cam, tx, ty, width, height, backgrounds, render_alphas, last_ids, v_render, v_alphas);
        topk[k] = wave_max_i(qs[k].binf);
        top = max(top, topk[k]);
    }
    top = min(top, e - 1);
    if (any_record && top >= s && lane == 0) *any_record = 1;
    if (top < s) return;

    float B[NP][16];
#pragma unroll
    for (int k = 0; k < NP; ++k) build_b_operands<CD>(B[k], k, lane, cam, tx, ty, width, height, v_render);
    MfmaWave<CD, ACC_STRIDE> mw;
    mw.init(&sh.w[wv][0][0], &sh.acc[wv][0][0], &sh.rowpos[wv][0], lane);
    float* const acc = mw.acc;
    // the batch accumulator starts cleared (and every flush leaves it cleared)
#pragma unroll
    for (int q = 0; q < ACC_STRIDE / 4; ++q)
        reinterpret_cast<float4*>(acc + lane * ACC_STRIDE)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float mcx = (float)(tx * MOBGS_TILE) + 8.f, mcy = (float)(ty * MOBGS_TILE) + 8.f;  // tile centre

    MFMA_T(t_prologue);
    MFMA_ACC(0, t_prologue - t_begin);
    for (int hi = top; hi >= s; hi -= 64) {
        const int n = min(64, hi - s + 1);
        MFMA_T(t_b0);
        // ---- stage: lane = entry hi - lane ------------------------------------------------------------------------
        const int idx = hi - lane;
        const bool valid = lane < n;
        const int g = valid ? flatten_ids[idx] : 0;
        const bool keep = valid && (!FILTER || cls.keeps(g));
        float4 r0 = make_float4(0.f, 0.f, 1.f, 0.f), r1 = make_float4(1.f, 1.f, 0.f, 0.f);
        int slot = 0;
        unsigned rm = 0u;
        if (valid) {
            const float4* r = reinterpret_cast<const float4*>(records + (size_t)g * RS);
            r0 = r[0];
            r1 = r[1];
            rm = isect_reach ? (unsigned)isect_reach[idx]
                 : cls.all_reach ? 0xFu
                                 : quadrant_reach_mask_rec(r0, r1, tx, ty);
            const TileRect tr = tile_rect(r0.x, r0.y, radii[g], tile_w, tile_h);
            slot = keep_index(keep_scan, cum_tiles[g] + (ty - tr.y0) * (tr.x1 - tr.x0) + (tx - tr.x0));
        }
        unsigned long long reach[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k)
            reach[k] = __builtin_amdgcn_ballot_w64(keep && ((rm >> k) & 1u) != 0u && idx <= topk[k]);
        MFMA_T(t_b1);
        MFMA_ACC(1, t_b1 - t_b0);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (reach[k] == 0ull) continue;
            walk_quadrant<CD, ACC_STRIDE, RS>(mw, records, g, reach[k], hi, qs[k].px, qs[k].py, qs[k].binf, qs[k].T,
                                              qs[k].behind, qs[k].tvab, qs[k].vo, B[k] MFMA_TARG);
        }
        // ---- flush: lane = entry --------------------------------------------------------------------------------
        MFMA_T(t_b2);
        MFMA_ACC(2, t_b2 - t_b1);
        wave_lds_fence();
        if (keep) {  // (class-restricted passes of the two classes share ONE slot buffer: never touch the other's)
            float a[16], out[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = reinterpret_cast<const float4*>(acc + lane * ACC_STRIDE)[q];
                reinterpret_cast<float4*>(acc + lane * ACC_STRIDE)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                a[4 * q] = v.x;
                a[4 * q + 1] = v.y;
                a[4 * q + 2] = v.z;
                a[4 * q + 3] = v.w;
            }
            convert_moments(a, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, mcx, mcy, out);
            float4* dst = reinterpret_cast<float4*>(grad_slots + (size_t)slot * RS);
#pragma unroll
            for (int q = 0; q < RQ; ++q) dst[q] = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
        }
        wave_lds_fence();
        MFMA_T(t_b3);
        MFMA_ACC(4, t_b3 - t_b2);
    }
#ifdef MOBGS_MFMA_TIMING
    t_acc[2] -= t_acc[3];
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_mfma_timing[i], t_acc[i]);
#endif
}

// =====================================================================================================================
// (2) team formulation: the four waves of a workgroup composite ONE tile, wave q its 8x8 quadrant q (one pixel per lane)
// =====================================================================================================================
// A wave's persistent state is a quarter of (1)'s -- 16 B operands, 10 cotangents, T / behind per pixel -- so four or
// five waves fit a SIMD, and the walk of a quadrant is what it is in (1).  What the team shares:
//   * staging: wave w stages batch positions [16 w, 16 w + 16) (flat id, record head, reach byte, gradient slot) in its
//     lanes 0..15, one batch AHEAD of the walk (the dependent loads id -> record / cum_tiles -> keep_scan are issued at
//     the top and at the end of the previous batch's walk), and publishes {id, reach} in LDS for the other waves;
//   * the sums: every wave adds into its own accumulator block; after a barrier wave w combines the four blocks of ITS
//     16 positions (lane = (position, source block): four 16-byte reads per lane, then the cross-lane sum over the four
//     blocks with v_permlane32_swap / v_permlane16_swap in fixed order), converts the moments and stores the records.
// Two barriers per batch of 64 entries.
struct TeamShared {
    float w[TILES_PER_WG][16][64] __attribute__((aligned(4096)));
    float acc[TILES_PER_WG][64][16] __attribute__((aligned(16)));
    int g_of[64];          // flat id of each batch position (-1: beyond the list / dropped by the class filter)
    unsigned rm_of[64];    // reach byte
    int rowpos[TILES_PER_WG][GROUP] __attribute__((aligned(16)));
    int top[TILES_PER_WG];
};

template <int CD, bool FILTER>
__device__ __forceinline__ void composite_bwd_team(
    int tile, int wv, int lane, TeamShared& sh, ClassSel cls, int tile_w, int tile_h, int width, int height,
    const float* __restrict__ records, const float* __restrict__ backgrounds, const int32_t* __restrict__ radii,
    const int32_t* __restrict__ cum_tiles, const int32_t* __restrict__ keep_scan,
    const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
    const float* __restrict__ render_alphas, const int32_t* __restrict__ last_ids, const float* __restrict__ v_render,
    const float* __restrict__ v_alphas, float* __restrict__ grad_slots, const uint8_t* __restrict__ isect_reach,
    int32_t* __restrict__ any_record) {
    static_assert(CD >= 1 && CD <= 10, "one accumulator block: 10 colour columns + 6 moment columns");
    constexpr int RS = (6 + CD + 3) & ~3;
    constexpr int RQ = RS / 4;
    const int quad = wv;
    const int tiles_per_cam = tile_w * tile_h;
    const int cam = tile / tiles_per_cam;
    const int tl = tile - cam * tiles_per_cam;
    const int ty = tl / tile_w, tx = tl - ty * tile_w;

    const int s = __builtin_amdgcn_readfirstlane(tile_offsets[tile]);
    const int e = __builtin_amdgcn_readfirstlane(tile_offsets[tile + 1]);
    if (e <= s) return;  // workgroup-uniform
#ifdef MOBGS_MFMA_TIMING
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 1, 0, 0};
#endif
    MFMA_T(t_begin);

    QuadState<CD> qs;
    qs.load(quad, lane, cam, tx, ty, width, height, backgrounds, render_alphas, last_ids, v_render, v_alphas);
    const int topq = wave_max_i(qs.binf);
    if (lane == 0) sh.top[wv] = topq;
    __syncthreads();
    int top = max(max(sh.top[0], sh.top[1]), max(sh.top[2], sh.top[3]));
    top = min(top, e - 1);
    if (any_record && top >= s && threadIdx.x == 0) *any_record = 1;
    if (top < s) return;  // workgroup-uniform

    float Bk[16];
    build_b_operands<CD>(Bk, quad, lane, cam, tx, ty, width, height, v_render);
    MfmaWave<CD, 16> mw;
    mw.init(&sh.w[wv][0][0], &sh.acc[wv][0][0], &sh.rowpos[wv][0], lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(mw.acc + lane * 16)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float mcx = (float)(tx * MOBGS_TILE) + 8.f, mcy = (float)(ty * MOBGS_TILE) + 8.f;

    // ---- staging pipeline of this wave: lanes 0..15 <-> batch positions 16 wv + lane ------------------------------------
    const bool stager = lane < 16;
    const int sp = 16 * wv + (lane & 15);
    auto load_id = [&](int hi_x) -> int {  // -1: no such entry
        const int idx = hi_x - sp;
        return (stager && hi_x >= s && idx >= s) ? flatten_ids[idx] : -1;
    };
    struct Head {
        float4 r0;
        float2 r1;
        int box;        // index of the (tile, splat) pair among the splat's bounding-box intersections
        unsigned rm;
    };
    auto load_head = [&](int g, int hi_x) -> Head {
        Head h;
        h.r0 = make_float4(0.f, 0.f, 1.f, 0.f);
        h.r1 = make_float2(1.f, 1.f);
        h.box = 0;
        h.rm = 0u;
        if (g >= 0) {
            const float4* r = reinterpret_cast<const float4*>(records + (size_t)g * RS);
            h.r0 = r[0];
            h.r1 = *reinterpret_cast<const float2*>(r + 1);
            h.rm = isect_reach ? (unsigned)isect_reach[hi_x - sp]
                   : cls.all_reach ? 0xFu
                                   : quadrant_reach_mask_rec(h.r0, make_float4(h.r1.x, h.r1.y, 0.f, 0.f), tx, ty);
            const TileRect tr = tile_rect(h.r0.x, h.r0.y, radii[g], tile_w, tile_h);
            h.box = cum_tiles[g] + (ty - tr.y0) * (tr.x1 - tr.x0) + (tx - tr.x0);
        }
        return h;
    };
    // prime: batch `top` completely, the ids of the next one
    // (ids of the other class of a class-restricted pass are dropped at once: their slots belong to the other pass)
    auto load_kept_id = [&](int hi_x) -> int {
        const int g = load_id(hi_x);
        return (g >= 0 && (!FILTER || cls.keeps(g))) ? g : -1;
    };
    int g_cur = load_kept_id(top);
    Head h_cur = load_head(g_cur, top);
    int slot_cur = g_cur >= 0 ? keep_index(keep_scan, h_cur.box) : 0;
    int g_nxt = load_kept_id(top - 64);
    if (stager) {
        sh.g_of[sp] = g_cur;
        sh.rm_of[sp] = h_cur.rm;
    }
    __syncthreads();

    MFMA_T(t_prologue);
    MFMA_ACC(0, t_prologue - t_begin);
    for (int hi = top; hi >= s; hi -= 64) {
        MFMA_T(t_b0);
        // next batch: record heads (their ids arrived during the previous batch), ids of the batch after
        const Head h_nxt = load_head(g_nxt, hi - 64);
        const int g_nn = load_kept_id(hi - 128);
        // this batch, lane = batch position
        const int gl = sh.g_of[lane];
        const unsigned rml = sh.rm_of[lane];
        const unsigned long long m =
            __builtin_amdgcn_ballot_w64(gl >= 0 && ((rml >> quad) & 1u) != 0u && (hi - lane) <= topq);
        MFMA_T(t_b1);
        MFMA_ACC(1, t_b1 - t_b0);
        if (m != 0ull)
            walk_quadrant<CD, 16, RS>(mw, records, gl, m, hi, qs.px, qs.py, qs.binf, qs.T, qs.behind, qs.tvab, qs.vo,
                                      Bk MFMA_TARG);
        // the gradient slots of the next batch (needs its cum_tiles: issued a whole walk ago)
        const int slot_nxt = g_nxt >= 0 ? keep_index(keep_scan, h_nxt.box) : 0;
        MFMA_T(t_b2);
        MFMA_ACC(2, t_b2 - t_b1);
        wave_lds_fence();
        __syncthreads();  // A: the four quadrant waves have added their sums of this batch; nobody reads g_of / rm_of
        {
            // lane = (position 16 wv + el, source block src): read, clear, then sum the four blocks across the lanes
            const int el = lane & 15, src = lane >> 4;
            float* ap = &sh.acc[src][16 * wv + el][0];
            float a[16], out[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = reinterpret_cast<const float4*>(ap)[q];
                reinterpret_cast<float4*>(ap)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                a[4 * q] = v.x;
                a[4 * q + 1] = v.y;
                a[4 * q + 2] = v.z;
                a[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {  // (block 0 + block 1) + (block 2 + block 3), the same in every lane
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[i]), false, false);
                const float x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                const auto q2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
                a[i] = __uint_as_float(q2[0]) + __uint_as_float(q2[1]);
            }
            if (stager && g_cur >= 0) {
                convert_moments(a, h_cur.r0.x, h_cur.r0.y, h_cur.r0.z, h_cur.r0.w, h_cur.r1.x, h_cur.r1.y, mcx, mcy, out);
                float4* dst = reinterpret_cast<float4*>(grad_slots + (size_t)slot_cur * RS);
#pragma unroll
                for (int q = 0; q < RQ; ++q) dst[q] = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
            }
            if (stager) {  // publish the next batch
                sh.g_of[sp] = g_nxt;
                sh.rm_of[sp] = h_nxt.rm;
            }
        }
        g_cur = g_nxt;
        h_cur = h_nxt;
        slot_cur = slot_nxt;
        g_nxt = g_nn;
        __syncthreads();  // B: accumulators cleared, next batch published
        MFMA_T(t_b3);
        MFMA_ACC(4, t_b3 - t_b2);
    }
#ifdef MOBGS_MFMA_TIMING
    t_acc[2] -= t_acc[3];
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_mfma_timing[i], t_acc[i]);
#endif
}

// TEAM_ALL: every scheduled tile is composited by the whole workgroup (the four slots of the workgroup one after the
// other); else only the tiles the schedule marks heavy, the others one wave each.
template <int CD, bool FILTER, bool TEAM_ALL>
__global__ void __launch_bounds__(64 * TILES_PER_WG) __attribute__((amdgpu_waves_per_eu(TEAM_ALL ? MOBGS_TEAM_WAVES : MOBGS_MFMA_WAVES)))
raster_bwd_mfma_kernel(int n_tiles_total, int n_groups, int tile_w, int tile_h, int width, int height,
                       const float* __restrict__ records, const float* __restrict__ backgrounds,
                       const int32_t* __restrict__ radii, const int32_t* __restrict__ cum_tiles,
                       const int32_t* __restrict__ keep_scan, const int32_t* __restrict__ tile_offsets,
                       const int32_t* __restrict__ flatten_ids, const float* __restrict__ render_alphas,
                       const int32_t* __restrict__ last_ids, const float* __restrict__ v_render,
                       const float* __restrict__ v_alphas, float* __restrict__ grad_slots,
                       const int32_t* __restrict__ tile_order, ClassSel cls, const uint8_t* __restrict__ isect_reach,
                       int32_t* __restrict__ any_record) {
    if (cls.gated_off()) return;  // every cotangent of this pass is zero (uniform over the launch)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if constexpr (TEAM_ALL) {
        __shared__ TeamShared sh;
        for (int t = 0; t < TILES_PER_WG; ++t) {
            const int slot = scheduled_tile(tile_order, n_groups, n_tiles_total, t);
            if (slot >= 0) {
                __syncthreads();  // the previous tile's LDS state (sh.top, accumulators) is no longer read
                composite_bwd_team<CD, FILTER>(slot & ~SCHED_HEAVY, wv, lane, sh, cls, tile_w, tile_h, width, height,
                                               records, backgrounds, radii, cum_tiles, keep_scan, tile_offsets,
                                               flatten_ids, render_alphas, last_ids, v_render, v_alphas, grad_slots,
                                               isect_reach, any_record);
                if (slot & SCHED_HEAVY) break;  // the four slots of a heavy workgroup name the same tile
            }
        }
    } else {
        __shared__ union {
            BwdMfmaShared one;
            TeamShared team;
        } sh;
        const int slot = scheduled_tile(tile_order, n_groups, n_tiles_total, wv);
        if (slot < 0) return;
        if (slot & SCHED_HEAVY)  // workgroup-uniform: all 4 slots of a heavy workgroup carry the flag
            composite_bwd_team<CD, FILTER>(slot & ~SCHED_HEAVY, wv, lane, sh.team, cls, tile_w, tile_h, width, height,
                                           records, backgrounds, radii, cum_tiles, keep_scan, tile_offsets, flatten_ids,
                                           render_alphas, last_ids, v_render, v_alphas, grad_slots, isect_reach,
                                           any_record);
        else
            composite_bwd_mfma<CD, FILTER>(slot, wv, lane, sh.one, cls, tile_w, tile_h, width, height, records,
                                           backgrounds, radii, cum_tiles, keep_scan, tile_offsets, flatten_ids,
                                           render_alphas, last_ids, v_render, v_alphas, grad_slots, isect_reach,
                                           any_record);
    }
}

// launcher used by mobgs_raster_bwd / mobgs_raster_class_bwd (raster.hip); false: channel count not built here.
// mode 1: one wave per tile (+ team for the schedule's heavy tiles), mode 2: team for every tile
bool raster_bwd_mfma_launch(int mode, int D, bool filter, int grid, hipStream_t st, int nt, int n_groups, int tile_w,
                            int tile_h, int width, int height, const float* records, const float* backgrounds,
                            const int32_t* radii, const int32_t* cum_tiles, const int32_t* keep_scan,
                            const int32_t* tile_offsets, const int32_t* flatten_ids, const float* render_alphas,
                            const int32_t* last_ids, const float* v_render, const float* v_alphas, float* grad_slots,
                            const int32_t* tile_order, ClassSel cls, const uint8_t* isect_reach, int32_t* any_record) {
#define MOBGS_LAUNCH_MFMA(CDV, FLT)                                                                                   \
    do {                                                                                                              \
        if (mode == 2)                                                                                                \
            hipLaunchKernelGGL((raster_bwd_mfma_kernel<CDV, FLT, true>), dim3(grid), dim3(64 * TILES_PER_WG), 0, st,  \
                               nt, n_groups, tile_w, tile_h, width, height, records, backgrounds, radii, cum_tiles,   \
                               keep_scan, tile_offsets, flatten_ids, render_alphas, last_ids, v_render, v_alphas,     \
                               grad_slots, tile_order, cls, isect_reach, any_record);                                 \
        else                                                                                                          \
            hipLaunchKernelGGL((raster_bwd_mfma_kernel<CDV, FLT, false>), dim3(grid), dim3(64 * TILES_PER_WG), 0, st, \
                               nt, n_groups, tile_w, tile_h, width, height, records, backgrounds, radii, cum_tiles,   \
                               keep_scan, tile_offsets, flatten_ids, render_alphas, last_ids, v_render, v_alphas,     \
                               grad_slots, tile_order, cls, isect_reach, any_record);                                 \
    } while (0)
    if (!filter) {
        switch (D) {
            case 1: MOBGS_LAUNCH_MFMA(1, false); return true;
            case 3: MOBGS_LAUNCH_MFMA(3, false); return true;
            case 4: MOBGS_LAUNCH_MFMA(4, false); return true;
            case 9: MOBGS_LAUNCH_MFMA(9, false); return true;
            case 10: MOBGS_LAUNCH_MFMA(10, false); return true;
            default: return false;
        }
    }
    switch (D) {
        case 1: MOBGS_LAUNCH_MFMA(1, true); return true;
        case 10: MOBGS_LAUNCH_MFMA(10, true); return true;
        default: return false;
    }
#undef MOBGS_LAUNCH_MFMA
}

}  // namespace mobgs

#ifdef MOBGS_MFMA_TIMING
extern "C" int mobgs_debug_mfma_timing(unsigned long long* out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(mobgs::g_mfma_timing), 8 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mobgs::g_mfma_timing), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
