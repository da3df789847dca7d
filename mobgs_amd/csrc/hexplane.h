// Shared by deform.hip (forward gather) and hexplane_bwd.hip (gradient scatter): plane table, bilinear tap, query
// normalisation of /root/reference/scene/hexplane.py:19-21,75-108,156-187.
#pragma once
#include "common.h"

namespace mobgs {

struct PlaneSet {
    const float* p[18];  // [level*6 + plane], channels-last [rb][ra][32]
    int ra[18];          // width  = resolution of the FIRST axis of the pair
    int rb[18];          // height = resolution of the SECOND axis of the pair
};
struct PlaneGradSet {
    float* p[18];
};

__constant__ const int kAxisA[6] = {0, 0, 0, 1, 1, 2};
__constant__ const int kAxisB[6] = {1, 2, 3, 2, 3, 3};

struct Tap {
    int o00, o01, o10, o11;  // element offsets of the 4 taps (channel 0)
    float wx, wy;            // fractional parts
    float gx, gy;            // d(ix)/d(coord), d(iy)/d(coord): 0 when the coordinate was clipped to the border
};

__device__ inline Tap make_tap(float x, float y, int ra, int rb) {
    // grid_sample, align_corners=True, padding_mode='border' (PyTorch clip_coordinates)
    Tap t;
    float ix = (x + 1.f) * 0.5f * (float)(ra - 1);
    float iy = (y + 1.f) * 0.5f * (float)(rb - 1);
    // clip_coordinates_set_grad: the borders themselves count as out of bounds (gradient 0 for ix <= 0, ix >= ra-1)
    t.gx = (ix > 0.f && ix < (float)(ra - 1)) ? 0.5f * (float)(ra - 1) : 0.f;
    t.gy = (iy > 0.f && iy < (float)(rb - 1)) ? 0.5f * (float)(rb - 1) : 0.f;
    ix = fminf(fmaxf(ix, 0.f), (float)(ra - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(rb - 1));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const int x1 = min(x0 + 1, ra - 1), y1 = min(y0 + 1, rb - 1);
    t.wx = ix - fx;
    t.wy = iy - fy;
    t.o00 = (y0 * ra + x0) * 32;
    t.o01 = (y0 * ra + x1) * 32;
    t.o10 = (y1 * ra + x0) * 32;
    t.o11 = (y1 * ra + x1) * 32;
    return t;
}

__device__ inline void normalized_query(const float* __restrict__ pts, const float* __restrict__ times,
                                        const float* __restrict__ aabb, int n, float q[4], float dq[3]) {
    // aabb[0] = xyz_max, aabb[1] = xyz_min (the reference's axis-inverted convention, hexplane.py:156-163)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a0 = aabb[k], a1 = aabb[3 + k];
        const float s = 2.0f / (a1 - a0);
        const float v = (pts[3 * n + k] - a0) * s - 1.0f;
        q[k] = fminf(fmaxf(v, -1.f), 1.f);
        dq[k] = (v >= -1.f && v <= 1.f) ? s : 0.f;
    }
    q[3] = times[n];
}

}  // namespace mobgs
