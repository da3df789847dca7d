// micro-benchmark: what does the cross-lane reduction of raster_bwd cost on gfx950, instruction by instruction?
// An ablated build (15 plain adds instead of wave_reduce16_scatter) takes 110 us off the 492 us kernel: ~205 SIMD-cycles
// per reduced list entry for 12 permlane swaps + 12 adds + 8 DPP adds + the hazard nops.  This counts shader cycles
// (s_memtime) per instruction for the swap instructions and for the whole sequence, at W waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o swap_cost swap_cost.hip && ./swap_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define NV 16

#define KERNEL(name, per_iter, body)                                                                          \
    __global__ void __launch_bounds__(512) k_##name(float* out, long long* cyc, int iters, float c0) {        \
        float v[NV];                                                                                          \
        for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 1e-3f + i + c0;                                      \
        __syncthreads();                                                                                      \
        const long long t0 = __builtin_readcyclecounter();                                                    \
        for (int it = 0; it < iters; ++it) {                                                                  \
            body                                                                                              \
        }                                                                                                     \
        const long long t1 = __builtin_readcyclecounter();                                                    \
        float s = 0;                                                                                          \
        for (int i = 0; i < NV; ++i) s += v[i];                                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                       \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;             \
    }                                                                                                         \
    static const int n_##name = per_iter;

#define ALL16 "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), \
              "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])

#define SWAP8(op)                                                                                             \
    asm volatile(op " %0, %8\n\t" op " %1, %9\n\t" op " %2, %10\n\t" op " %3, %11\n\t"                        \
                 op " %4, %12\n\t" op " %5, %13\n\t" op " %6, %14\n\t" op " %7, %15" : ALL16);

// 8 swaps per iteration, back to back (different registers: independent)
KERNEL(swap32, 8, SWAP8("v_permlane32_swap_b32"))
KERNEL(swap16, 8, SWAP8("v_permlane16_swap_b32"))
// the same 8 "instructions" as plain full-rate moves, for the baseline of this loop shape
KERNEL(mov8, 8, SWAP8("v_mov_b32"))
// 8 DPP adds under a bank mask (the halving steps inside a row)
#define DPP8(ctl)                                                                                             \
    asm volatile("v_add_f32_dpp %0, %8, %8 " ctl "\n\tv_add_f32_dpp %1, %9, %9 " ctl "\n\t"                  \
                 "v_add_f32_dpp %2, %10, %10 " ctl "\n\tv_add_f32_dpp %3, %11, %11 " ctl "\n\t"              \
                 "v_add_f32_dpp %4, %12, %12 " ctl "\n\tv_add_f32_dpp %5, %13, %13 " ctl "\n\t"              \
                 "v_add_f32_dpp %6, %14, %14 " ctl "\n\tv_add_f32_dpp %7, %15, %15 " ctl : ALL16);
KERNEL(dpp_ror8, 8, DPP8("row_ror:8 row_mask:0xf bank_mask:0xc"))
KERNEL(dpp_quad, 8, DPP8("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
// DPP row_bcast31 (GCN wave-level broadcast of lane 31 into rows 2, 3) and wave_ror (whole-wave rotate by one lane)
KERNEL(dpp_bcast31, 8, DPP8("row_bcast:31 row_mask:0xc bank_mask:0xf"))
KERNEL(dpp_wave_ror, 8, DPP8("wave_ror:1 row_mask:0xf bank_mask:0xf"))
// ds_swizzle: swap with lane ^ 16 inside every 32 lanes, on the LDS crossbar (no memory access, not a VALU instruction)
#define SWZ8                                                                                                  \
    asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,16)\n\tds_swizzle_b32 %1, %1 offset:swizzle(SWAP,16)\n\t" \
                 "ds_swizzle_b32 %2, %2 offset:swizzle(SWAP,16)\n\tds_swizzle_b32 %3, %3 offset:swizzle(SWAP,16)\n\t" \
                 "ds_swizzle_b32 %4, %4 offset:swizzle(SWAP,16)\n\tds_swizzle_b32 %5, %5 offset:swizzle(SWAP,16)\n\t" \
                 "ds_swizzle_b32 %6, %6 offset:swizzle(SWAP,16)\n\tds_swizzle_b32 %7, %7 offset:swizzle(SWAP,16)\n\t" \
                 "s_waitcnt lgkmcnt(0)" : ALL16);
KERNEL(swizzle16, 8, SWZ8)

// the whole reduction of raster_bwd (12 swaps, 12 adds, 8 DPP adds, nops as the hazards require), once per iteration
__device__ __forceinline__ float reduce16(float (&v)[NV]) {
    asm volatile("s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %2, %10\n\t"
                 "v_permlane32_swap_b32 %3, %11\n\tv_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\t"
                 "v_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15\n\ts_nop 1" : ALL16);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += v[i + 8];
    asm volatile("s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\t"
                 "v_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += v[i + 4];
    float u0, u1, w;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(u0) : "v"(v[0]));
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\ts_nop 1" : "+v"(u0) : "v"(v[2]));
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(u1) : "v"(v[1]));
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\ts_nop 1" : "+v"(u1) : "v"(v[3]));
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(w) : "v"(u0));
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\ts_nop 1" : "+v"(w) : "v"(u1));
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1" : "+v"(w));
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1" : "+v"(w));
    return w;
}
// the same sums with the DPP tail as ONE block: the two independent chains interleaved, a nop only where a DPP read
// follows the write of its source by fewer than two instructions
__device__ __forceinline__ float reduce16_v2(float (&v)[NV]) {
    asm volatile("s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %2, %10\n\t"
                 "v_permlane32_swap_b32 %3, %11\n\tv_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\t"
                 "v_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15\n\ts_nop 1" : ALL16);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += v[i + 8];
    asm volatile("s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\t"
                 "v_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += v[i + 4];
    float u0, u1, w;
    asm volatile("s_nop 1\n\t"
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
                 : "=&v"(u0), "=&v"(u1), "=&v"(w) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    return w;
}
KERNEL(reduce16_v2, 1, { const float w = reduce16_v2(v); _Pragma("unroll") for (int i = 0; i < NV; ++i) v[i] = w + (float)i; })
KERNEL(reduce16, 1, { const float w = reduce16(v); _Pragma("unroll") for (int i = 0; i < NV; ++i) v[i] = w + (float)i; })
// the loop shape alone (16 adds that re-seed the registers)
KERNEL(reseed, 1, { const float w = v[0] + v[5]; _Pragma("unroll") for (int i = 0; i < NV; ++i) v[i] = w + (float)i; })

template <typename K>
void run(const char* name, K kern, int per_iter, int waves_per_simd) {
    const int threads = 256 * (waves_per_simd < 2 ? 1 : 2);
    const int blocks_per_cu = waves_per_simd / (threads / 256);
    const int blocks = 256 * (blocks_per_cu < 1 ? 1 : blocks_per_cu);
    const int iters = 4096;
    float* out;
    long long* cyc;
    const int n_waves = blocks * threads / 64;
    (void)hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
    (void)hipMalloc(&cyc, (size_t)n_waves * sizeof(long long));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 64, 1.0001f);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0001f);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(n_waves);
    (void)hipMemcpy(h.data(), cyc, n_waves * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[n_waves / 2];
    printf("%-12s W=%d  cycles per iteration per wave %.1f   SIMD-cycles per instruction (or per sequence) %.2f\n", name,
           waves_per_simd, med / iters, med / ((double)iters * per_iter * waves_per_simd));
    (void)hipFree(out);
    (void)hipFree(cyc);
}

#define RUN(name, w) run(#name, k_##name, n_##name, w)
int main() {
    for (int w : {1, 2, 4}) {
        RUN(mov8, w);
        RUN(swap32, w);
        RUN(swap16, w);
        RUN(dpp_ror8, w);
        RUN(dpp_quad, w);
        RUN(dpp_bcast31, w);
        RUN(dpp_wave_ror, w);
        RUN(swizzle16, w);
        RUN(reseed, w);
        RUN(reduce16, w);
        RUN(reduce16_v2, w);
    }
    return 0;
}
