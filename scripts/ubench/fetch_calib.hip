// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the compositors.
// MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access
// widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".
//   stream_read   : 16 B per lane, fully coalesced, 1 GiB (beyond the 256 MiB Infinity Cache)
//   gather64      : 64-byte records picked at random from a 1 GiB table, 4 lanes x 16 B per record (the splat-record
//                   gather of raster_fwd / raster_bwd), 16 Mi records = 1 GiB of useful bytes
//   gather64_small: the same from a 19 MiB table (= 300 k splat records: L2 / Infinity-Cache resident)
//   write64       : one 64-byte record per 16 lanes into every SECOND 64-byte half of a zero-filled 1 GiB buffer
//                   (the gradient-slot store of raster_bwd: do partial-line writes cost a read?)
//   write128      : full 128-byte lines
// usage: rocprofv3 --pmc FETCH_SIZE ... -- ./fetch_calib ; rocprofv3 --pmc WRITE_SIZE ... -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void stream_read(const float4* __restrict__ src, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

__device__ inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ void gather64(const float4* __restrict__ table, uint32_t n_records_mask, size_t n_gathers, float* out) {
    float acc = 0.f;
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const int part = threadIdx.x & 3;
    for (size_t r = t >> 2; r < n_gathers; r += ((size_t)gridDim.x * blockDim.x) >> 2) {
        const uint32_t rec = hash32((uint32_t)r) & n_records_mask;
        const float4 v = table[(size_t)rec * 4 + part];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int STRIDE_F4>  // records of 4 float4 written every STRIDE_F4 float4 (4 = dense 64 B, 8 = every second half line)
__global__ void write_records(float4* __restrict__ dst, size_t n_records) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const int part = threadIdx.x & 3;
    for (size_t r = t >> 2; r < n_records; r += ((size_t)gridDim.x * blockDim.x) >> 2)
        dst[r * STRIDE_F4 + part] = make_float4((float)r, 1.f, 2.f, 3.f);
}

int main() {
    const size_t GiB = 1ull << 30;
    float4 *big, *small;
    float* out;
    (void)hipMalloc(&big, GiB);
    (void)hipMalloc(&small, 32ull << 20);
    (void)hipMalloc(&out, 64);
    (void)hipMemset(big, 0, GiB);
    (void)hipMemset(small, 0, 32ull << 20);
    const int grid = 256 * 16, block = 256;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream_read, dim3(grid), dim3(block), 0, 0, big, GiB / 16, out);
        hipLaunchKernelGGL(gather64, dim3(grid), dim3(block), 0, 0, big, (uint32_t)(GiB / 64 - 1), (size_t)(GiB / 64), out);
        hipLaunchKernelGGL(gather64, dim3(grid), dim3(block), 0, 0, small, (uint32_t)((16ull << 20) / 64 - 1),
                           (size_t)(GiB / 64), out);
        (void)hipMemsetAsync(big, 0, GiB, 0);
        hipLaunchKernelGGL(write_records<8>, dim3(grid), dim3(block), 0, 0, big, (size_t)(GiB / 128));
        (void)hipMemsetAsync(big, 0, GiB, 0);
        hipLaunchKernelGGL(write_records<4>, dim3(grid), dim3(block), 0, 0, big, (size_t)(GiB / 128));
    }
    (void)hipDeviceSynchronize();
    printf("useful bytes per launch: stream_read %zu, gather64 %zu (both tables), write_records<8> %zu (into %zu of lines), "
           "write_records<4> %zu\n", GiB, GiB, GiB / 2, GiB, GiB / 2);
    return 0;
}
