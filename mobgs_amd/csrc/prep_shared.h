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
    // no implicit contraction here: this body is compiled into several kernels that must agree to the bit, and the
    // reference's torch expressions round every product and sum on their own (plain operators, NOT the __fmul_rn /
    // __fadd_rn spellings: those are inline functions of the toolchain's header with contraction allowed inside, which
    // this pragma does not reach -- measured: u = fma(t, n - 1, -idx) in the ISA)
#pragma clang fp contract(off)
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
    const float omu2 = omu * omu, u2 = u * u;
    H.h00 = (1.f + 2.f * u) * omu2;
    H.h10 = u * omu2;
    H.h01 = u2 * (3.f - 2.f * u);
    H.h11 = u2 * (u - 1.f);
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
    // no implicit contraction here: this body is compiled into several kernels that must agree to the bit, and the
    // reference's torch expressions round every product and sum on their own (plain operators, NOT the __fmul_rn /
    // __fadd_rn spellings: those are inline functions of the toolchain's header with contraction allowed inside, which
    // this pragma does not reach -- measured: u = fma(t, n - 1, -idx) in the ISA)
#pragma clang fp contract(off)
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
            // (as the reference's torch expression evaluates it: every product rounded, the sums left to right)
            m[k] = (H.h00 * p1 + H.h10 * m0 + H.h01 * p2 + H.h11 * m1) * 1e-2f;
            s[k] = expf(ldf(in.d_scaling, 3 * (size_t)j + k));
            col[6 + k] = tfp * ldf(in.d_ft, 3 * (size_t)j + k);
        }
        const float4 r = ld4(in.d_rotation, j);
        const float4 w = ld4(in.d_omega, j);
        // (rotation + t * omega as torch evaluates it: product rounded, then the sum -- no contraction in this body)
        q[0] = r.x + tfp * w.x; q[1] = r.y + tfp * w.y; q[2] = r.z + tfp * w.z; q[3] = r.w + tfp * w.w;
        o = 1.f / (1.f + expf(-ldf(in.d_opacity, j)));
#pragma unroll
        for (int k = 0; k < 6; ++k) col[k] = ldf(in.d_fdc, 6 * (size_t)j + k);
    }
}

// ---- backward: one splat's leaf gradients from the cotangents of its state, all in registers ----------------------
// vm / vq / vc: cotangents of position / UN-normalised rotation / the 9 colour features; vs: cotangent of the scales
// TIMES the scales (d exp = exp); vo: cotangent of the opacity times o (1 - o).  ACC: add to what the buffers hold
// (ops.LeafGradSink), else overwrite (the control-point rows must have been cleared: prep_bwd_clear_rows).
template <bool ACC, typename G>
__device__ __forceinline__ void prep_bwd_apply(
    int i, int Ns, const float* __restrict__ times, const long long* __restrict__ d_ncp,
    const float* __restrict__ d_trbf, const float (&vm)[3], const float (&vq)[4], const float (&vs)[3], float vo,
    const float (&vc)[9], float* __restrict__ g_s_xyz, G* __restrict__ g_s_scaling, G* __restrict__ g_s_rotation,
    G* __restrict__ g_s_opacity, G* __restrict__ g_s_fdc, G* __restrict__ g_s_ft, float* __restrict__ g_d_control,
    G* __restrict__ g_d_scaling, G* __restrict__ g_d_rotation, G* __restrict__ g_d_omega, G* __restrict__ g_d_opacity,
    G* __restrict__ g_d_fdc, G* __restrict__ g_d_ft) {
#pragma clang fp contract(off)
    if (i < Ns) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            g_s_xyz[3 * i + k] = (ACC ? g_s_xyz[3 * i + k] : 0.f) + vm[k];
            stf(g_s_scaling, 3 * (size_t)i + k, (ACC ? ldf(g_s_scaling, 3 * (size_t)i + k) : 0.f) + vs[k]);
            stf(g_s_ft, 3 * (size_t)i + k, (ACC ? ldf(g_s_ft, 3 * (size_t)i + k) : 0.f) + 0.0f * vc[6 + k]);
        }
        {
            float4 q = make_float4(vq[0], vq[1], vq[2], vq[3]);
            if (ACC) {
                const float4 o = ld4(g_s_rotation, i);
                q = make_float4(o.x + q.x, o.y + q.y, o.z + q.z, o.w + q.w);
            }
            st4(g_s_rotation, i, q);
        }
        stf(g_s_opacity, i, (ACC ? ldf(g_s_opacity, i) : 0.f) + vo);
#pragma unroll
        for (int k = 0; k < 6; ++k) stf(g_s_fdc, 6 * (size_t)i + k, (ACC ? ldf(g_s_fdc, 6 * (size_t)i + k) : 0.f) + vc[k]);
    } else {
        const int j = i - Ns;
        const float tfp = times[0] - d_trbf[j];
        const int n = (int)d_ncp[j];
        const Hermite H = hermite_setup(times[1], n);
        float* gc = g_d_control + (size_t)j * 36;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = vm[k] * 1e-2f;
            float a0 = 0.f, a1 = H.h00 * v, a2 = H.h01 * v, a3 = 0.f;
            const float vm0 = H.h10 * v, vm1 = H.h11 * v;
            if (H.left_edge) {
                a2 += vm0;
                a1 -= vm0;
            } else {
                a2 += 0.5f * vm0;
                a0 -= 0.5f * vm0;
            }
            if (H.right_edge) {
                a2 += vm1;
                a1 -= vm1;
            } else {
                a3 += 0.5f * vm1;
                a1 -= 0.5f * vm1;
            }
            // knots coincide only at the curve ends (i0 == i1: a0 is 0; i3 == i2: a3 is 0), where the dead term is
            // simply not stored.  !ACC: the rows were zero-filled by the kernel, plain stores; ACC: read-modify-write
            if (ACC) {
                if (!H.left_edge) gc[3 * H.i0 + k] += a0;
                gc[3 * H.i1 + k] += a1;
                gc[3 * H.i2 + k] += a2;
                if (!H.right_edge) gc[3 * H.i3 + k] += a3;
            } else {
                if (!H.left_edge) gc[3 * H.i0 + k] = a0;
                gc[3 * H.i1 + k] = a1;
                gc[3 * H.i2 + k] = a2;
                if (!H.right_edge) gc[3 * H.i3 + k] = a3;
            }
            stf(g_d_scaling, 3 * (size_t)j + k, (ACC ? ldf(g_d_scaling, 3 * (size_t)j + k) : 0.f) + vs[k]);
            stf(g_d_ft, 3 * (size_t)j + k, (ACC ? ldf(g_d_ft, 3 * (size_t)j + k) : 0.f) + tfp * vc[6 + k]);
        }
        {
            float4 q = make_float4(vq[0], vq[1], vq[2], vq[3]);
            float4 w = make_float4(tfp * vq[0], tfp * vq[1], tfp * vq[2], tfp * vq[3]);
            if (ACC) {
                const float4 o = ld4(g_d_rotation, j);
                const float4 p = ld4(g_d_omega, j);
                q = make_float4(o.x + q.x, o.y + q.y, o.z + q.z, o.w + q.w);
                w = make_float4(p.x + w.x, p.y + w.y, p.z + w.z, p.w + w.w);
            }
            st4(g_d_rotation, j, q);
            st4(g_d_omega, j, w);
        }
        stf(g_d_opacity, j, (ACC ? ldf(g_d_opacity, j) : 0.f) + vo);
#pragma unroll
        for (int k = 0; k < 6; ++k) stf(g_d_fdc, 6 * (size_t)j + k, (ACC ? ldf(g_d_fdc, 6 * (size_t)j + k) : 0.f) + vc[k]);
    }
}

// !ACC: the 144-byte control-point gradient rows of this WAVE's dynamic splats are contiguous: clear them with coalesced
// 16-byte stores (a thread clearing its own row issues 36 stores that each touch 64 lines), then every thread drops its
// <= 12 non-zero entries into its row.  Same wave, program order: the fill's stores are complete (s_waitcnt vmcnt(0) of
// the wavefront-scope release) before the entries are written.  Call with ALL threads of the wave, before any return.
__device__ __forceinline__ void prep_bwd_clear_rows(int Ns, int Nd, float* __restrict__ g_d_control) {
    const int wave_first = (blockIdx.x * blockDim.x + (threadIdx.x & ~63)) - Ns;  // first dynamic index
    const int j0 = max(wave_first, 0), j1 = min(wave_first + 64, Nd);
    if (j1 > j0) {
        float4* row = reinterpret_cast<float4*>(g_d_control + (size_t)j0 * 36);
        const int n4 = (j1 - j0) * 9;
        for (int t = threadIdx.x & 63; t < n4; t += 64) row[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0);
    }
}

}  // namespace mobgs
