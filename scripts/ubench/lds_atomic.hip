// micro-benchmark: throughput of LDS float atomics (ds_add_f32, no return) against plain LDS stores and integer
// atomics, 16 waves per CU, conflict-free addresses (lane -> bank) and the worst case (all 16 waves on the same cells).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
    __shared__ float tab[16 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += 1024) tab[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int cell = (it * 8 + u) & 15;
            if (MODE == 0) atomicAdd(&tab[(wv * 16 + cell) * 64 + lane], v);                 // private rows per wave
            if (MODE == 1) atomicAdd(&tab[cell * 64 + lane], v);                               // all waves, same 16 rows
            if (MODE == 2) tab[(wv * 16 + cell) * 64 + lane] = v;                              // plain store
            if (MODE == 3) atomicAdd(reinterpret_cast<int*>(&tab[(wv * 16 + cell) * 64 + lane]), 1);  // int atomic
        }
    }
    __syncthreads();
    out[blockIdx.x * 1024 + threadIdx.x] = tab[threadIdx.x];
}
template <int MODE>
void run(const char* name) {
    float* out;
    (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 4096;
    k<MODE><<<256, 1024>>>(out, 16);
    (void)hipEventRecord(e0);
    k<MODE><<<256, 1024>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = 16.0 * iters * 8;
    printf("%-28s %.3f ms  -> %.1f ns per wave-instruction per CU (%.1f cycles at 2.1 GHz)\n", name, ms,
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.1);
}
int main() {
    run<0>("ds_add_f32 private rows");
    run<1>("ds_add_f32 shared rows");
    run<2>("ds_write_b32");
    run<3>("ds_add_u32 private rows");
    return 0;
}
