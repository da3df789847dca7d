// Normals from a depth map (replaces /root/reference/main_utils.py:95-141 get_normals, called once per view and
// iteration by train.py:590): back-project every pixel with its view direction, n = normalize(cross(right - left,
// top - bottom)) on the interior, zeros on the 1-pixel border.  The reference rebuilds the [H,W,3] direction map
// with numpy and copies it to the device on every call; here the direction comes from the five intrinsics in
// registers.  Backward is a gather (each pixel revisits the four centre pixels it is a neighbour of): no atomics,
// bit-reproducible.
#include "common.h"

namespace mobgs {

struct NormalCam {
    float fx, fy, cx, cy, skew, offset;
};

__device__ __forceinline__ void view_dir(const NormalCam& c, int i, int j, float d[3]) {
    const float y = (((float)i + c.offset) - c.cy) / c.fy;
    const float x = (((float)j + c.offset) - c.cx - y * c.skew) / c.fx;
    d[0] = x;
    d[1] = y;
    d[2] = 1.f;
}

// un-normalised normal of centre pixel (i, j) (interior) and the two difference vectors
__device__ __forceinline__ void centre_vectors(const NormalCam& c, const float* __restrict__ z, int W, int i, int j,
                                               float l2r[3], float b2t[3], float n[3]) {
    float dr[3], dl[3], dt[3], db[3];
    view_dir(c, i, j + 1, dr);
    view_dir(c, i, j - 1, dl);
    view_dir(c, i - 1, j, dt);
    view_dir(c, i + 1, j, db);
    const float zr = z[(size_t)i * W + j + 1], zl = z[(size_t)i * W + j - 1];
    const float zt = z[(size_t)(i - 1) * W + j], zb = z[(size_t)(i + 1) * W + j];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        l2r[k] = dr[k] * zr - dl[k] * zl;
        b2t[k] = dt[k] * zt - db[k] * zb;
    }
    n[0] = l2r[1] * b2t[2] - l2r[2] * b2t[1];
    n[1] = l2r[2] * b2t[0] - l2r[0] * b2t[2];
    n[2] = l2r[0] * b2t[1] - l2r[1] * b2t[0];
}

__global__ void __launch_bounds__(256) normals_fwd_kernel(int H, int W, NormalCam c, const float* __restrict__ z,
                                                            float* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int i = p / W, j = p - i * W;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (i >= 1 && i <= H - 2 && j >= 1 && j <= W - 2) {
        float l2r[3], b2t[3], n[3];
        centre_vectors(c, z, W, i, j, l2r, b2t, n);
        const float len = fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-12f);  // F.normalize eps
        nx = n[0] / len;
        ny = n[1] / len;
        nz = n[2] / len;
    }
    const size_t P = (size_t)H * W;
    out[p] = nx;
    out[P + p] = ny;
    out[2 * P + p] = nz;
}

// gradient of centre (i, j) w.r.t. its two difference vectors, given the cotangent of its unit normal
__device__ __forceinline__ void centre_grads(const NormalCam& c, const float* __restrict__ z,
                                             const float* __restrict__ g, int H, int W, int i, int j, float v_l2r[3],
                                             float v_b2t[3]) {
    float l2r[3], b2t[3], n[3];
    centre_vectors(c, z, W, i, j, l2r, b2t, n);
    const size_t P = (size_t)H * W, q = (size_t)i * W + j;
    const float gv[3] = {g[q], g[P + q], g[2 * P + q]};
    const float raw = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    float vn[3];
    if (raw > 1e-12f) {
        const float inv = 1.f / raw;
        const float u[3] = {n[0] * inv, n[1] * inv, n[2] * inv};
        const float dot = u[0] * gv[0] + u[1] * gv[1] + u[2] * gv[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) vn[k] = (gv[k] - u[k] * dot) * inv;
    } else {  // clamped denominator: n / 1e-12
#pragma unroll
        for (int k = 0; k < 3; ++k) vn[k] = gv[k] * 1e12f;
    }
    // n = l2r x b2t  ->  v_l2r = b2t x vn,  v_b2t = vn x l2r
    v_l2r[0] = b2t[1] * vn[2] - b2t[2] * vn[1];
    v_l2r[1] = b2t[2] * vn[0] - b2t[0] * vn[2];
    v_l2r[2] = b2t[0] * vn[1] - b2t[1] * vn[0];
    v_b2t[0] = vn[1] * l2r[2] - vn[2] * l2r[1];
    v_b2t[1] = vn[2] * l2r[0] - vn[0] * l2r[2];
    v_b2t[2] = vn[0] * l2r[1] - vn[1] * l2r[0];
}

__global__ void __launch_bounds__(256) normals_bwd_kernel(int H, int W, NormalCam c, const float* __restrict__ z,
                                                            const float* __restrict__ g, float* __restrict__ v_z) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int i = p / W, j = p - i * W;
    auto interior = [&](int a, int b) { return a >= 1 && a <= H - 2 && b >= 1 && b <= W - 2; };
    float vp[3] = {0.f, 0.f, 0.f};  // cotangent of the back-projected point of pixel p
    float a[3], b[3];
    if (interior(i, j - 1)) {  // p is the RIGHT point of the centre to its left
        centre_grads(c, z, g, H, W, i, j - 1, a, b);
        vp[0] += a[0]; vp[1] += a[1]; vp[2] += a[2];
    }
    if (interior(i, j + 1)) {  // LEFT point of the centre to its right
        centre_grads(c, z, g, H, W, i, j + 1, a, b);
        vp[0] -= a[0]; vp[1] -= a[1]; vp[2] -= a[2];
    }
    if (interior(i + 1, j)) {  // TOP point of the centre below
        centre_grads(c, z, g, H, W, i + 1, j, a, b);
        vp[0] += b[0]; vp[1] += b[1]; vp[2] += b[2];
    }
    if (interior(i - 1, j)) {  // BOTTOM point of the centre above
        centre_grads(c, z, g, H, W, i - 1, j, a, b);
        vp[0] -= b[0]; vp[1] -= b[1]; vp[2] -= b[2];
    }
    float d[3];
    view_dir(c, i, j, d);
    v_z[p] = d[0] * vp[0] + d[1] * vp[1] + d[2] * vp[2];
}

}  // namespace mobgs

using namespace mobgs;

extern "C" {

int mobgs_normals_fwd(int H, int W, float fx, float fy, float cx, float cy, float skew, float pixel_offset,
                      const float* z, float* normals, void* stream) {
    if (H < 1 || W < 1 || !z || !normals) {
        set_error("mobgs_normals_fwd: bad arguments H=%d W=%d", H, W);
        return MOBGS_E_INVALID;
    }
    const NormalCam c = {fx, fy, cx, cy, skew, pixel_offset};
    hipLaunchKernelGGL(normals_fwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, c, z,
                       normals);
    return check_launch("normals_fwd_kernel");
}

int mobgs_normals_bwd(int H, int W, float fx, float fy, float cx, float cy, float skew, float pixel_offset,
                      const float* z, const float* v_normals, float* v_z, void* stream) {
    if (H < 1 || W < 1 || !z || !v_normals || !v_z) {
        set_error("mobgs_normals_bwd: bad arguments H=%d W=%d", H, W);
        return MOBGS_E_INVALID;
    }
    const NormalCam c = {fx, fy, cx, cy, skew, pixel_offset};
    hipLaunchKernelGGL(normals_bwd_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, c, z,
                       v_normals, v_z);
    return check_launch("normals_bwd_kernel");
}

}  // extern "C"
