// Shared helpers for the libmobgs_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mobgs_hip.h"

#define MOBGS_WAVE 64

namespace mobgs {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MOBGS_E_LAUNCH;
    }
    return MOBGS_OK;
}

__host__ __device__ inline int record_stride(int channels) { return (6 + channels + 3) & ~3; }

// XCD-aware remap of a linear workgroup id.  The dispatcher places workgroup b on XCD b % 8
// (MI355X_MICROARCH.md, "Workgroup dispatch"); giving each XCD one contiguous chunk of the logical
// index space keeps neighbouring tiles (which share splats) behind the same 4 MiB L2.  Only a
// performance hint: any placement gives the same result.
__device__ inline int xcd_chunked(int b, int n) {
    const int per = (n + 7) >> 3;
    const int logical = (b & 7) * per + (b >> 3);
    return logical;  // may be >= n for the ragged tail; callers bounds-check
}

// Tile rectangle [x0,x1) x [y0,y1) touched by a splat (gsplat isect_tiles.cu [upstream]).
struct TileRect {
    int x0, y0, x1, y1;
};
__host__ __device__ inline TileRect tile_rect(float mx, float my, int radius, int tile_w, int tile_h) {
    const float inv = 1.0f / (float)MOBGS_TILE;
    const float tr = (float)radius * inv;
    const float tx = mx * inv, ty = my * inv;
    TileRect r;
    float fx0 = floorf(tx - tr), fx1 = ceilf(tx + tr), fy0 = floorf(ty - tr), fy1 = ceilf(ty + tr);
    // clamp in float first so huge coordinates cannot overflow the int conversion
    fx0 = fminf(fmaxf(fx0, 0.f), (float)tile_w);
    fx1 = fminf(fmaxf(fx1, 0.f), (float)tile_w);
    fy0 = fminf(fmaxf(fy0, 0.f), (float)tile_h);
    fy1 = fminf(fmaxf(fy1, 0.f), (float)tile_h);
    r.x0 = (int)fx0;
    r.x1 = (int)fx1;
    r.y0 = (int)fy0;
    r.y1 = (int)fy1;
    return r;
}

}  // namespace mobgs
