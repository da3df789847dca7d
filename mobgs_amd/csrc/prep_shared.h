// Per-splat state build shared by prep.hip (the stand-alone kernels) and project.hip (the projection kernel that builds
// the state itself, mobgs_prep_project_and_bin_fused): loads that widen half attributes, the Hermite basis, and the
// forward evaluation of ONE splat.  See prep.hip's header for the reference lines this restates.
#pragma once
#include <hip/hip_fp16.h>

#include "common.h"

namespace mobgs {

__device__ inline float ldf(const float* p, size_t i) { return p[i]; }
__device__ inline float ldf(const __half* p, size_t i) { return __half2float(p[i]); }
__device__ inline float4 ld4(const float* p, size_t i) { return reinterpret_cast<const float4*>(p)[i]; }
__device__ inline float4 ld4(const __half* p, size_t i) {
    const uint2 u = reinterpret_cast<const uint2*>(p)[i];  // 4 halves = 8 bytes
    const __half2 a = *reinterpret_cast<const __half2*>(&u.x), b = *reinterpret_cast<const __half2*>(&u.y);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    return make_float4(fa.x, fa.y, fb.x, fb.y);
}
__device__ inline void stf(float* p, size_t i, float v) { p[i] = v; }
__device__ inline void stf(__half* p, size_t i, float v) { p[i] = __float2half(v); }
__device__ inline void st4(float* p, size_t i, float4 v) { reinterpret_cast<float4*>(p)[i] = v; }
__device__ inline void st4(__half* p, size_t i, float4 v) {
    const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
    uint2 u;
    u.x = *reinterpret_cast<const unsigned*>(&a);
    u.y = *reinterpret_cast<const unsigned*>(&b);
    reinterpret_cast<uint2*>(p)[i] = u;
}

struct Hermite {
    int i0, i1, i2, i3;  // left, index, right, right-right knots
    float h00, h10, h01, h11;
    bool left_edge, right_edge;
};

__device__ inline Hermite hermite_setup(float t, int n) {
    Hermite H;
    const float ts = t * (float)(n - 1);
    int idx = (int)floorf(ts);
    idx = min(max(idx, 0), n - 2);
    H.i1 = idx;
    H.i0 = min(max(idx - 1, 0), n - 1);
    H.i2 = min(max(idx + 1, 0), n - 1);
    H.i3 = min(max(idx + 2, 0), n - 1);
    const float u = ts - (float)idx;
    const float omu = 1.f - u;
    H.h00 = (1.f + 2.f * u) * (omu * omu);
    H.h10 = u * (omu * omu);
    H.h01 = (u * u) * (3.f - 2.f * u);
    H.h11 = (u * u) * (u - 1.f);
    H.left_edge = (H.i0 == H.i1);
    H.right_edge = (H.i3 == H.i2);
    return H;
}

// raw (pre-activation) parameters of the two sets, as mobgs_prep_fwd takes them
template <typename A>
struct PrepIn {
    int Ns, Nd;
    const float* times;  // {t_feat, t_curve}
    const float* s_xyz;
    const A *s_scaling, *s_rotation, *s_opacity, *s_fdc, *s_ft;
    const float* d_control;
    const long long* d_ncp;
    const A *d_scaling, *d_rotation, *d_omega, *d_opacity, *d_fdc, *d_ft;
    const float* d_trbf;
};

// splat i of the concatenated set -> position, UN-normalised rotation, scales, opacity, 9 colour features
template <typename A>
__device__ __forceinline__ void prep_splat(const PrepIn<A>& in, int i, float (&m)[3], float (&q)[4], float (&s)[3],
                                           float& o, float (&col)[9]) {
    if (i < in.Ns) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            m[k] = in.s_xyz[3 * i + k];
            s[k] = expf(ldf(in.s_scaling, 3 * (size_t)i + k));  // (accurate exp: the scale decides the integer radius)
            col[6 + k] = 0.0f * ldf(in.s_ft, 3 * (size_t)i + k);
        }
        // the reference normalises static rotations (get_rotation_stat); emit them raw, see prep.hip's header
        const float4 r = ld4(in.s_rotation, i);
        q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w;
        o = 1.f / (1.f + expf(-ldf(in.s_opacity, i)));
#pragma unroll
        for (int k = 0; k < 6; ++k) col[k] = ldf(in.s_fdc, 6 * (size_t)i + k);
    } else {
        const int j = i - in.Ns;
        const float t_feat = in.times[0], t_curve = in.times[1];
        const float tfp = t_feat - in.d_trbf[j];
        const int n = (int)in.d_ncp[j];
        const Hermite H = hermite_setup(t_curve, n);
        const float* cp = in.d_control + (size_t)j * 36;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float p0 = cp[3 * H.i0 + k], p1 = cp[3 * H.i1 + k], p2 = cp[3 * H.i2 + k], p3 = cp[3 * H.i3 + k];
            const float m0 = H.left_edge ? (p2 - p1) : (p2 - p0) * 0.5f;
            const float m1 = H.right_edge ? (p2 - p1) : (p3 - p1) * 0.5f;
            m[k] = (H.h00 * p1 + H.h10 * m0 + H.h01 * p2 + H.h11 * m1) * 1e-2f;
            s[k] = expf(ldf(in.d_scaling, 3 * (size_t)j + k));
            col[6 + k] = tfp * ldf(in.d_ft, 3 * (size_t)j + k);
        }
        const float4 r = ld4(in.d_rotation, j);
        const float4 w = ld4(in.d_omega, j);
        // (explicit FMAs: this function is compiled into two kernels, which must round alike)
        q[0] = __fmaf_rn(tfp, w.x, r.x); q[1] = __fmaf_rn(tfp, w.y, r.y);
        q[2] = __fmaf_rn(tfp, w.z, r.z); q[3] = __fmaf_rn(tfp, w.w, r.w);
        o = 1.f / (1.f + expf(-ldf(in.d_opacity, j)));
#pragma unroll
        for (int k = 0; k < 6; ++k) col[k] = ldf(in.d_fdc, 6 * (size_t)j + k);
    }
}

}  // namespace mobgs
