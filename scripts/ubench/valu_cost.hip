// micro-benchmark: issue cost of individual VALU instructions on gfx950 (wave64), SIMD-cycles per instruction at an
// assumed 2.4 GHz.  16 independent chains per thread, 8 waves per SIMD: throughput, not latency.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define OP(name, asmtext)                                                                    \
    __global__ void __launch_bounds__(256) k_##name(float* out, int iters, float c0) {       \
        float a[16];                                                                          \
        for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;                          \
        float b = c0, c = c0 * 0.5f;                                                          \
        for (int it = 0; it < iters; ++it) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(asmtext : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                     \
        float s = 0;                                                                          \
        for (int i = 0; i < 16; ++i) s += a[i];                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                       \
    }
OP(fma, "v_fma_f32 %0, %0, %1, %2")
OP(mul, "v_mul_f32 %0, %0, %1")
OP(add, "v_add_f32 %0, %0, %1")
OP(min, "v_min_f32 %0, %0, %1")
OP(max, "v_max_f32 %0, %0, %1")
OP(med3, "v_med3_f32 %0, %0, %1, %2")
OP(mov, "v_mov_b32 %0, %1")
OP(cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc")
OP(cmp, "v_cmp_lt_f32 vcc, %0, %1")
OP(cmp_sgpr, "v_cmp_lt_f32 s[10:11], %0, %1")
OP(cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
OP(exp, "v_exp_f32 %0, %0")
OP(rcp, "v_rcp_f32 %0, %0")
OP(and_b32, "v_and_b32 %0, %0, %1")
OP(add_u32, "v_add_u32 %0, %0, %1")
OP(max_u32, "v_max_u32 %0, %0, %1")
OP(fmac_e32, "v_fmac_f32 %0, %1, %2")
OP(sub_e32, "v_sub_f32 %0, %0, %1")
OP(cvt, "v_cvt_f32_u32 %0, %0")
template <typename K>
void run(const char* name, K kern, double per_iter) {
    float* out;
    const int blocks = 256 * 8, iters = 4096;
    (void)hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out, 16, 1.0001f);
    (void)hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, iters, 1.0001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters * per_iter;
    const double cyc = ms * 1e-3 * 2.4e9 * 1024;
    printf("%-24s %8.3f ms  %.2f SIMD-cycles per wave64 instruction (at 2.4 GHz)\n", name, ms, cyc / wave_instr);
    (void)hipFree(out);
}
int main() {
#define R(n, c) run(#n, k_##n, c)
    R(fma, 16); R(fmac_e32, 16); R(mul, 16); R(add, 16); R(sub_e32, 16); R(min, 16); R(max, 16); R(med3, 16); R(mov, 16);
    R(cndmask_vcc, 16); R(cmp, 16); R(cmp_sgpr, 16); R(cmp_cnd, 32); R(exp, 16); R(rcp, 16); R(and_b32, 16);
    R(add_u32, 16); R(max_u32, 16); R(cvt, 16);
    return 0;
}
