// micro-benchmark: cycles per v_permlane32_swap / v_permlane16_swap / bank-masked DPP add on gfx950 (the cross-lane
// instructions of raster_bwd's gradient reduction), s_memtime around long independent streams,
// 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o swap_cost swap_cost.hip && ./swap_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CHAINS 16
#define UNROLL 8
#define KERNEL2(name, body)                                                                                   \
    __global__ void __launch_bounds__(512) k_##name(float* out, long long* cyc, int iters, float c0) {        \
        float a[CHAINS];                                                                                      \
        for (int i = 0; i < CHAINS; ++i) a[i] = threadIdx.x * 1e-3f + i;                                      \
        __syncthreads();                                                                                      \
        const long long t0 = __builtin_readcyclecounter();                                                    \
        for (int it = 0; it < iters; ++it) {                                                                  \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                              \
                _Pragma("unroll") for (int i = 0; i < CHAINS; i += 2)                                         \
                    asm volatile(body : "+v"(a[i]), "+v"(a[i + 1]));                                          \
            }                                                                                                 \
        }                                                                                                     \
        const long long t1 = __builtin_readcyclecounter();                                                    \
        float s = 0;                                                                                          \
        for (int i = 0; i < CHAINS; ++i) s += a[i];                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                       \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;             \
    }
KERNEL2(swap32, "v_permlane32_swap_b32 %0, %1")
KERNEL2(swap16, "v_permlane16_swap_b32 %0, %1")
KERNEL2(dpp_masked, "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc")
KERNEL2(dpp_full, "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf")
KERNEL2(add, "v_add_f32 %0, %0, %1")
KERNEL2(swap32_add, "v_permlane32_swap_b32 %0, %1\n\tv_add_f32 %0, %0, %1")
template <typename K>
void run(const char* name, K kern, int per_asm) {
    const int waves_per_simd = 4, threads = 512, blocks = 512, iters = 2048;
    float* out; long long* cyc;
    const int n_waves = blocks * threads / 64;
    (void)hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
    (void)hipMalloc(&cyc, (size_t)n_waves * sizeof(long long));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 64, 1.0001f);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0001f);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(n_waves);
    (void)hipMemcpy(h.data(), cyc, n_waves * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double n_asm = (double)iters * UNROLL * (CHAINS / 2);
    printf("%-12s cycles per asm statement per SIMD (4 waves): median %.2f  (%d instruction%s per statement)\n", name,
           (double)h[n_waves / 2] / (n_asm * waves_per_simd), per_asm, per_asm > 1 ? "s" : "");
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run("swap32", k_swap32, 1);
    run("swap16", k_swap16, 1);
    run("dpp_masked", k_dpp_masked, 1);
    run("dpp_full", k_dpp_full, 1);
    run("add", k_add, 1);
    run("swap32+add", k_swap32_add, 2);
    return 0;
}
