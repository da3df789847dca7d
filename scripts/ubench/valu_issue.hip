// micro-benchmark: how many SIMD cycles does one wave64 VALU instruction occupy on gfx950?
//
// MI355X_MICROARCH.md says 2 (SIMD-32: 32 lanes per cycle; 157.3 TFLOP/s fp32 vector peak = 64 FLOP/clk/SIMD); round 1's
// scripts/ubench/valu_cost.hip reported 3.0-3.5 -- but it converted WALL time into cycles at an assumed 2.4 GHz, and a
// chip running nothing but fp32 FMAs on every SIMD does not hold 2.4 GHz.  This one counts cycles with s_memtime
// (the shader clock) around the instruction stream of every wave and reports, for W waves per SIMD:
//     cycles per wave64 instruction per SIMD = (cycles the slowest wave of a SIMD needed) / (W x instructions per wave)
// together with the effective clock (shader cycles / wall time) the kernel ran at.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CHAINS 16
#define UNROLL 8   // instructions per chain per loop iteration

#define KERNEL(name, body)                                                                                   \
    __global__ void __launch_bounds__(512) k_##name(float* out, long long* cyc, int iters, float c0) {        \
        float a[CHAINS];                                                                                      \
        for (int i = 0; i < CHAINS; ++i) a[i] = threadIdx.x * 1e-3f + i;                                      \
        float b = c0, c = c0 * 0.5f;                                                                          \
        __syncthreads();                                                                                      \
        const long long t0 = __builtin_readcyclecounter();                                                    \
        for (int it = 0; it < iters; ++it) {                                                                  \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                              \
                _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(body : "+v"(a[i]) : "v"(b), "v"(c)); \
            }                                                                                                 \
        }                                                                                                     \
        const long long t1 = __builtin_readcyclecounter();                                                    \
        float s = 0;                                                                                          \
        for (int i = 0; i < CHAINS; ++i) s += a[i];                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                       \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;             \
    }

KERNEL(fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(mul, "v_mul_f32 %0, %0, %1")
KERNEL(add, "v_add_f32 %0, %0, %1")
KERNEL(mov, "v_mov_b32 %0, %1")
KERNEL(max, "v_max_f32 %0, %0, %1")
KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(cnd_sgpr, "v_cndmask_b32 %0, %0, %1, s[10:11]")
KERNEL(cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL(cmp_sgpr, "v_cmp_lt_f32 s[10:11], %0, %1")
KERNEL(min, "v_min_f32 %0, %0, %1")
KERNEL(med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(and_b32, "v_and_b32 %0, %0, %1")
KERNEL(add_u32, "v_add_u32 %0, %0, %1")
KERNEL(lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL(cvt, "v_cvt_f32_i32 %0, %0")
KERNEL(fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(mul_neg, "v_mul_f32 %0, -%0, %1")
KERNEL(dpp_add, "v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(readlane, "v_readlane_b32 s12, %0, 3")
KERNEL(exp, "v_exp_f32 %0, %0")
KERNEL(rcp, "v_rcp_f32 %0, %0")
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(512) k_pk_fma(float* out, long long* cyc, int iters, float c0) {
    v2f a[CHAINS];
    for (int i = 0; i < CHAINS; ++i) a[i] = v2f{threadIdx.x * 1e-3f + i, (float)i};
    v2f b = {c0, c0}, c = {c0 * 0.5f, c0 * 0.25f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < CHAINS; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// 64-bit moves: would two registers per instruction halve the cost of clearing an accumulator block?
#define KERNEL64(name, body)                                                                                  \
    __global__ void __launch_bounds__(512) k_##name(float* out, long long* cyc, int iters, float c0) {        \
        v2f a[CHAINS];                                                                                        \
        for (int i = 0; i < CHAINS; ++i) a[i] = v2f{threadIdx.x * 1e-3f + i, (float)i};                       \
        v2f b = {c0, c0 * 2.f};                                                                               \
        __syncthreads();                                                                                      \
        const long long t0 = __builtin_readcyclecounter();                                                    \
        for (int it = 0; it < iters; ++it) {                                                                  \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                              \
                _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(body : "+v"(a[i]) : "v"(b));  \
            }                                                                                                 \
        }                                                                                                     \
        const long long t1 = __builtin_readcyclecounter();                                                    \
        float s = 0;                                                                                          \
        for (int i = 0; i < CHAINS; ++i) s += a[i].x + a[i].y;                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                       \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;             \
    }
KERNEL64(mov_b64, "v_mov_b64 %0, %1")
KERNEL64(pk_mov, "v_pk_mov_b32 %0, %1, %1")
KERNEL64(pk_add, "v_pk_add_f32 %0, %0, %1")

template <typename K>
void run(const char* name, K kern, int waves_per_simd, bool packed) {
    // one workgroup per CU with 4 * W waves (<= 8 per SIMD would need 2048 threads: use W <= 2 per block and more blocks)
    const int threads = 256 * (waves_per_simd < 2 ? 1 : 2);                 // 4 or 8 waves per block
    const int blocks_per_cu = waves_per_simd / (threads / 256);             // blocks resident per CU
    const int blocks = 256 * (blocks_per_cu < 1 ? 1 : blocks_per_cu);
    const int iters = 2048;
    float* out;
    long long* cyc;
    const int n_waves = blocks * threads / 64;
    (void)hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
    (void)hipMalloc(&cyc, (size_t)n_waves * sizeof(long long));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 64, 1.0001f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(n_waves);
    (void)hipMemcpy(h.data(), cyc, n_waves * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[n_waves / 2], mx = (double)h[n_waves - 1];
    const double instr_per_wave = (double)iters * UNROLL * CHAINS;
    const double cpi = med / (instr_per_wave * waves_per_simd);
    const double cpi_max = mx / (instr_per_wave * waves_per_simd);
    // readcyclecounter on gfx9 = s_memtime: constant 100 MHz reference clock on this part if it is NOT the shader
    // clock -- print both interpretations' inputs: ticks and wall time
    const double wall_cyc_24 = ms * 1e-3 * 2.4e9;
    printf("%-8s W=%d  ticks(median)=%.0f ticks(max)=%.0f wall=%.3f ms  ticks/wall = %.1f MHz | cycles/instr/SIMD: by "
           "ticks %.3f, by wall@2.4GHz %.3f%s\n", name, waves_per_simd, med, mx, ms, med / (ms * 1e3), cpi,
           wall_cyc_24 / (instr_per_wave * waves_per_simd), packed ? " (two FMAs per lane)" : "");
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run("fma", k_fma, w, false);
    }
    for (int w : {4}) {
        run("cnd_sgpr", k_cnd_sgpr, w, false);
        run("cmp_vcc", k_cmp_vcc, w, false);
        run("cmp_sgpr", k_cmp_sgpr, w, false);
        run("min", k_min, w, false);
        run("med3", k_med3, w, false);
        run("and_b32", k_and_b32, w, false);
        run("add_u32", k_add_u32, w, false);
        run("lshl", k_lshl, w, false);
        run("cvt", k_cvt, w, false);
        run("fmac", k_fmac, w, false);
        run("mul_neg", k_mul_neg, w, false);
        run("dpp_add", k_dpp_add, w, false);
        run("readlane", k_readlane, w, false);
    }
    for (int w : {2, 4}) {
        run("mul", k_mul, w, false);
        run("add", k_add, w, false);
        run("mov", k_mov, w, false);
        run("max", k_max, w, false);
        run("cndmask", k_cndmask, w, false);
        run("exp", k_exp, w, false);
        run("rcp", k_rcp, w, false);
        run("pk_fma", k_pk_fma, w, true);
        run("mov_b64", k_mov_b64, w, true);
        run("pk_mov", k_pk_mov, w, true);
        run("pk_add", k_pk_add, w, true);
    }
    return 0;
}
