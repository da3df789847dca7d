// micro-benchmark: issue rate of plain vs packed fp32 FMA and v_exp on gfx950 (used to size the VALU floor in DESIGN.md)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2_ __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a[16];
    float2_ p[8];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 8; ++i) p[i] = float2_{a[2 * i], a[2 * i + 1]};
    const float m = 1.0001f, c = 1e-3f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __fmaf_rn(a[i], m, c);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = p[i] * float2_{m, m} + float2_{c, c};
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_exp2f(a[i] * 1e-3f);
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = (a[i] > c) ? a[i] * m : c;  // cmp + cndmask + mul
        } else if (MODE == 4) {  // 8 x v_mov_b64 (is a 64-bit move one pass or two?)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mov_b64 %0, %1" : "=v"(p[i]) : "v"(p[(i + 1) & 7]));
        } else {                 // 16 x v_mov_b32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, double ops_per_iter_per_thread) {
    float* out;
    const int blocks = 256 * 8, iters = 4096;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 16);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters * ops_per_iter_per_thread;  // wave-level instructions
    const double cyc = ms * 1e-3 * 2.4e9 * 1024;                                    // SIMD-cycles at 2.4 GHz
    printf("%-28s %8.3f ms  %.2f SIMD-cycles per wave64 instruction (at 2.4 GHz)\n", name, ms, cyc / wave_instr);
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32", 16);
    run<1>("v_pk_fma_f32", 8);
    run<2>("v_mul + v_exp_f32", 32);
    run<3>("v_cmp + v_cndmask + v_mul", 48);
    run<4>("v_mov_b64", 8);
    run<5>("v_mov_b32", 16);
    return 0;
}
