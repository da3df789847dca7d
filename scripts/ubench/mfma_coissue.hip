// micro-benchmark (round 4): can the gradient sums of raster_bwd ride on the matrix pipe beside the per-pixel VALU work?
//
// One "group" of the planned kernel = 8 (entry, quadrant) evaluations on the VALU (per evaluation ~46 full-rate ops, 8
// half-rate selects / compares, 2 transcendentals, 2 ds_write_b32 of the pair weights) followed by the group's
// reduction over the 64 pixels of the quadrant: 4 ds_read_b128 of the transposed weights + 16 v_mfma_f32_16x16x4_f32
// (rows = 8 entries x {fac, v_sigma}, K = 4 pixels per instruction, columns = 10 colour cotangents + 6 moments).
// Variants, each timed with s_memtime for W = 1..4 waves per SIMD:
//   valu      the 8 evaluations alone                          -> VALU cycles per group
//   mfma      the 16 MFMAs alone (two accumulator chains)      -> matrix-pipe cycles per group (expected 16 x 32 = 512)
//   burst     8 evaluations, then the 16 MFMAs back to back    -> what an in-order wave pays; overlap only across waves
//   spread    2 MFMAs of the PREVIOUS group after every evaluation (software pipeline inside one wave)
// Output: cycles per group per SIMD = slowest wave of a SIMD / groups, x (1 / W) for the per-wave-normalised figure, and
// the overlap the two pipes reached: (valu + mfma - combined) / min(valu, mfma).
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue mfma_coissue.hip && ./mfma_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CH 16
// one evaluation: 46 A (fma), 8 B (cndmask / cmp / min), 2 C (exp, rcp), 2 LDS stores
#define EVAL(a, b, c, waddr)                                                                                   \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < CH; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
        _Pragma("unroll") for (int i = 0; i < CH; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
        _Pragma("unroll") for (int i = 0; i < 14; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));     \
        _Pragma("unroll") for (int i = 4; i < 8; ++i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));              \
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[8]));                                                         \
        asm volatile("v_rcp_f32 %0, %0" : "+v"(a[9]));                                                         \
        if (LDS) {                                                                                             \
            asm volatile("ds_write_b32 %0, %1" ::"v"(waddr), "v"(a[10]) : "memory");                           \
            asm volatile("ds_write_b32 %0, %1 offset:2048" ::"v"(waddr), "v"(a[11]) : "memory");               \
        }                                                                                                      \
    }

// MODE 0 valu, 1 mfma, 2 burst, 3 spread; 4-6: the same with v_mfma_f32_16x16x32_bf16 (weights and cotangents split into
// bf16 hi + lo parts: 8 instructions per group -- [hi|lo] x [b_hi|b_hi] and [hi|lo] x [b_lo|0] for each 16-pixel slice)
template <int MODE, bool LDS>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc, int groups, float c0) {
    __shared__ float w[16][1024 + 64];
    float a[CH];
    for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 1e-3f + i;
    float b = c0, c = c0 * 0.5f;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 1024; i += 64) w[wv][i] = 0.f;
    const unsigned waddr = (unsigned)(size_t)(&w[wv][lane]);
    const unsigned raddr = (unsigned)(size_t)(&w[wv][(lane & 15) * 64 + ((((lane >> 4) * 4) ^ ((lane + 4) & 15)) << 2)]);
    f4 A[4], acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float B[16];
    for (int i = 0; i < 16; ++i) B[i] = c0 + i + lane;
    for (int i = 0; i < 4; ++i) A[i] = f4{c0, c0, c0, c0};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int g = 0; g < groups; ++g) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) EVAL(a, b, c, waddr)
        }
        if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                EVAL(a, b, c, waddr)
                // (asm volatile: the scheduler otherwise hoists all sixteen into the first evaluations)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(A[r >> 1][(2 * r) & 3]), "v"(B[2 * r]));
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(A[r >> 1][(2 * r + 1) & 3]), "v"(B[2 * r + 1]));
            }
        }
        if (MODE == 5) {
#pragma unroll
            for (int r = 0; r < 8; ++r) EVAL(a, b, c, waddr)
        }
        if (MODE == 6) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                EVAL(a, b, c, waddr)
                if (r & 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(A[r >> 1]), "v"(A[(r >> 1) ^ 1]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(A[r >> 1]), "v"(A[(r >> 1) ^ 1]));
            }
        }
        if (MODE == 4 || MODE == 5) {
#pragma unroll
            for (int m = 0; m < 8; m += 2) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(A[m >> 1]), "v"(A[(m >> 1) ^ 1]));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(A[m >> 1]), "v"(A[(m >> 1) ^ 1]));
            }
        }
        if (MODE != 0 && LDS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 4; ++t)
                asm volatile("ds_read_b128 %0, %1" : "=v"(A[t]) : "v"(raddr ^ (unsigned)(t << 4)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3])::"memory");
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int m = 0; m < 16; m += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m >> 2][m & 3], B[m], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m >> 2][(m + 1) & 3], B[m + 1], acc1, 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < CH; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE, bool LDS>
static double run(int waves_per_simd, int groups, float* d_out, long long* d_cyc, double* ghz) {
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = ncu;                   // one workgroup of 256 * W threads per CU: W waves on every SIMD
    const int threads = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, LDS>), dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 8, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, LDS>), dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, groups, 1.0001f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4 * waves_per_simd);
    hipMemcpy(h.data(), d_cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    *ghz = med / (ms * 1e6);
    return med / groups;  // cycles per group as seen by a wave sharing its SIMD with (waves_per_simd - 1) others
}

int main() {
    float* d_out;
    long long* d_cyc;
    hipMalloc(&d_out, 256 * 4 * 256 * 8 * sizeof(float));
    hipMalloc(&d_cyc, 256 * 4 * 8 * sizeof(long long));
    const int groups = 2000;
    printf("cycles per group (8 evaluations + 16 MFMA 16x16x4 f32) seen by one wave; per SIMD = that / waves\n");
    for (int lds = 0; lds < 2; ++lds) {
        printf("%s\n", lds ? "-- with the LDS traffic (2 ds_write_b32 per evaluation, 4 ds_read_b128 per group)" : "-- registers only");
        printf("%6s %10s %10s %10s %10s %9s %9s %6s | %8s %8s %8s\n", "waves", "valu", "mfma", "burst", "spread", "ovl_burst", "ovl_sprd", "GHz", "bf16", "bf_burst", "bf_sprd");
        for (int w = 1; w <= 4; ++w) {
            double g0, g1, g2, g3;
            double v, m, bu, sp, b4, b5, b6;
            if (lds) {
                v = run<0, true>(w, groups, d_out, d_cyc, &g0);
                m = run<1, true>(w, groups, d_out, d_cyc, &g1);
                bu = run<2, true>(w, groups, d_out, d_cyc, &g2);
                sp = run<3, true>(w, groups, d_out, d_cyc, &g3);
                b4 = run<4, true>(w, groups, d_out, d_cyc, &g3);
                b5 = run<5, true>(w, groups, d_out, d_cyc, &g3);
                b6 = run<6, true>(w, groups, d_out, d_cyc, &g3);
            } else {
                v = run<0, false>(w, groups, d_out, d_cyc, &g0);
                m = run<1, false>(w, groups, d_out, d_cyc, &g1);
                bu = run<2, false>(w, groups, d_out, d_cyc, &g2);
                sp = run<3, false>(w, groups, d_out, d_cyc, &g3);
                b4 = run<4, false>(w, groups, d_out, d_cyc, &g3);
                b5 = run<5, false>(w, groups, d_out, d_cyc, &g3);
                b6 = run<6, false>(w, groups, d_out, d_cyc, &g3);
            }
            printf("%6d %10.0f %10.0f %10.0f %10.0f %9.2f %9.2f %6.2f | %8.0f %8.0f %8.0f\n", w, v, m, bu, sp,
                   (v + m - bu) / std::min(v, m), (v + m - sp) / std::min(v, m), g2, b4, b5, b6);
        }
    }
    return 0;
}
