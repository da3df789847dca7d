// HexPlane feature gather + deformation MLP for gfx950 (the API the reference exposes as `deform_network`).
//
// Restates for the GPU:
//   /root/reference/scene/hexplane.py:19-21,75-108,165-187   normalize_aabb, 3 levels x 6 planes bilinear
//                                                            grid_sample(align_corners=True, border), product
//                                                            over planes, concat over levels -> 96 features
//   /root/reference/scene/deformation.py:56-73,158-199       Linear(96,128); three heads ReLU-Linear(128,128)-
//                                                            ReLU-Linear(128,{7,3,4}); point / scale / rotation update
//   /root/reference/scene/deformation.py:417-438             quat2mat on [1 | dx[3:7]] / 5-norm
//   /root/reference/utils/graphics_utils.py:117-140          batch_quaternion_multiply (normalised)
//
// hexplane_fwd : 32 lanes = the 32 channels of one tap (planes are passed channels-LAST, so a bilinear tap is one
//                128-byte row); 2 points per wave instruction; the 35 MB of planes live in L2 / Infinity Cache.
// The MLP heads + update rules (the only dense contraction on the path -> fp32 MFMA) live in deform_bwd.hip, the
// gradient of this gather in hexplane_bwd.hip.
#include <atomic>

#include "common.h"
#include "hexplane.h"

namespace mobgs {

__global__ void __launch_bounds__(256)
hexplane_fwd_kernel(int N, const float* __restrict__ pts, const float* __restrict__ times,
                    const float* __restrict__ aabb, PlaneSet planes, float* __restrict__ feat) {
    const int c = threadIdx.x & 31;
    const int half = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int nhalf = (gridDim.x * blockDim.x) >> 5;
    for (int n = half; n < N; n += nhalf) {
        float q[4], dq[3];
        normalized_query(pts, times, aabb, n, q, dq);
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            float prod = 1.f;
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const int id = l * 6 + p;
                const Tap t = make_tap(q[kAxisA[p]], q[kAxisB[p]], planes.ra[id], planes.rb[id]);
                const float* g = planes.p[id] + c;
                const float v00 = g[t.o00], v01 = g[t.o01], v10 = g[t.o10], v11 = g[t.o11];
                const float s = (v00 * (1.f - t.wx) + v01 * t.wx) * (1.f - t.wy) +
                                (v10 * (1.f - t.wx) + v11 * t.wx) * t.wy;
                prod *= s;
            }
            feat[(size_t)n * 96 + l * 32 + c] = prod;
        }
    }
}

}  // namespace mobgs

using namespace mobgs;

extern "C" {

static int fill_planes(PlaneSet& ps, const float* const* planes_host, const int32_t* ra_host,
                       const int32_t* rb_host) {
    for (int i = 0; i < 18; ++i) {
        ps.p[i] = planes_host[i];
        ps.ra[i] = ra_host[i];
        ps.rb[i] = rb_host[i];
        if (!ps.p[i] || ps.ra[i] < 1 || ps.rb[i] < 1) return MOBGS_E_INVALID;
    }
    return MOBGS_OK;
}

int mobgs_hexplane_fwd(int N, const float* pts, const float* times, const float* aabb,
                       const float* const* planes_host, const int32_t* ra_host, const int32_t* rb_host, float* feat,
                       void* stream) {
    PlaneSet ps;
    if (N < 0 || fill_planes(ps, planes_host, ra_host, rb_host) != MOBGS_OK) {
        set_error("mobgs_hexplane_fwd: bad arguments");
        return MOBGS_E_INVALID;
    }
    if (N == 0) return MOBGS_OK;
    int grid = (N + 7) / 8;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(hexplane_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, pts, times, aabb, ps,
                       feat);
    return check_launch("hexplane_fwd_kernel");
}

}  // extern "C"
