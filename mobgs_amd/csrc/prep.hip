// Per-splat state build for MoBGS's render(): one streaming kernel instead of ~60 tiny torch launches.
//
// Restates, for gfx950, the per-Gaussian glue of the reference's render()/get_flow():
//   /root/reference/gaussian_renderer/__init__.py:23-56    interpolate_cubic_hermite (per-splat knot count)
//   /root/reference/gaussian_renderer/__init__.py:93-125   time offset, rotation = _rotation + tfp*_omega,
//                                                          scales = exp, colours = [f_dc | tfp * f_t]
//   /root/reference/scene/gaussian_model.py:209-254        get_scaling / get_opacity / get_features(_static)
//   /root/reference/gaussian_renderer/__init__.py:181-185  cat(static, dynamic)
//
// Output rows [0,Ns) are the static splats, [Ns,Ns+Nd) the dynamic ones -- already concatenated, in the
// operator-level layout (means [N,3], quats [N,4] wxyz, scales [N,3], opacities [N], colours [N,9]).
// Quaternions are emitted UN-normalised: the projection kernel normalises (as upstream gsplat does), and
// normalise(normalise(q)) == normalise(q) including its Jacobian, so the reference's extra F.normalize is a no-op.
//
// times[0] = t_feat  = time + delta/max_time (unclamped; drives tfp = t_feat - trbf_center)
// times[1] = t_curve = clamp(t_feat, 0, 1)    (drives the Hermite spline)
// They are read from DEVICE memory so a BLCE exposure offset living on the GPU never forces a host sync.
//
// Attribute storage (BASELINE config #5, "fp16 Gaussian attributes"): the kernels are templated on the storage type
// A of the per-splat ATTRIBUTES (scaling, rotation, omega, opacity, features_dc, features_t): float, or IEEE half
// read straight from HBM and widened in registers (all arithmetic stays fp32; 80 -> 46 bytes per static splat,
// 232 -> 188 per dynamic one).  Positions / spline control points / time centres stay fp32: a half cannot hold a
// position in centimetres.  The backward kernel writes the attribute gradients in the type G the caller asks for
// (half leaves need half .grad tensors; the multi-render accumulation buffers of LeafGradSink stay fp32).
#include "prep_shared.h"

namespace mobgs {

template <typename A>
__global__ void __launch_bounds__(256)
prep_fwd_kernel(PrepIn<A> in,
                // out
                float* __restrict__ means, float* __restrict__ quats, float* __restrict__ scales,
                float* __restrict__ opac, float* __restrict__ colors) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = in.Ns + in.Nd;
    if (i >= N) return;
    // several time instants in one launch (grid.y; the K sub-frames of a blurry view): instant kk reads times[kk] and
    // writes row block kk of means / quats / colors; scales and opacities do not depend on time and are written once
    const int kk = blockIdx.y;
    in.times += 2 * kk;
    means += (size_t)kk * N * 3;
    quats += (size_t)kk * N * 4;
    colors += (size_t)kk * N * 9;
    float m[3], q[4], s[3], o, col[9];
    prep_splat(in, i, m, q, s, o, col);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        means[3 * i + k] = m[k];
        if (kk == 0) scales[3 * i + k] = s[k];
    }
    reinterpret_cast<float4*>(quats)[i] = make_float4(q[0], q[1], q[2], q[3]);
    if (kk == 0) opac[i] = o;
#pragma unroll
    for (int k = 0; k < 9; ++k) colors[9 * i + k] = col[k];
}

// ACC: the leaf gradients are ADDED to what the buffers hold (several renders of one backward pass write into one
// set of buffers, see mobgs_amd.ops.LeafGradSink) instead of overwriting them.
// One time instant of the backward pass for splat i.  FIRST: the cotangents of the time-independent outputs (scales,
// opacities) are taken in (they exist once, not per instant).
template <bool ACC, typename G>
__device__ __forceinline__ void prep_bwd_instant(
    int i, int Ns, const float* __restrict__ times, const long long* __restrict__ d_ncp,
    const float* __restrict__ d_trbf, const float* __restrict__ scales, const float* __restrict__ opac,
    const float* __restrict__ v_means, const float* __restrict__ v_quats, const float* __restrict__ v_scales,
    const float* __restrict__ v_opac, const float* __restrict__ v_colors, float* __restrict__ g_s_xyz,
    G* __restrict__ g_s_scaling, G* __restrict__ g_s_rotation, G* __restrict__ g_s_opacity, G* __restrict__ g_s_fdc,
    G* __restrict__ g_s_ft, float* __restrict__ g_d_control, G* __restrict__ g_d_scaling, G* __restrict__ g_d_rotation,
    G* __restrict__ g_d_omega, G* __restrict__ g_d_opacity, G* __restrict__ g_d_fdc, G* __restrict__ g_d_ft) {
    float vm[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f}, vo = 0.f, vc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) vc[k] = v_colors ? v_colors[9 * i + k] : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (v_means) vm[k] = v_means[3 * i + k];
        if (v_scales) vs[k] = v_scales[3 * i + k] * scales[3 * i + k];  // d exp = exp
    }
    if (v_quats) {
        const float4 r = reinterpret_cast<const float4*>(v_quats)[i];
        vq[0] = r.x; vq[1] = r.y; vq[2] = r.z; vq[3] = r.w;
    }
    if (v_opac) {
        const float o = opac[i];
        vo = v_opac[i] * o * (1.f - o);
    }
    prep_bwd_apply<ACC, G>(i, Ns, times, d_ncp, d_trbf, vm, vq, vs, vo, vc, g_s_xyz, g_s_scaling, g_s_rotation, g_s_opacity,
                           g_s_fdc, g_s_ft, g_d_control, g_d_scaling, g_d_rotation, g_d_omega, g_d_opacity, g_d_fdc, g_d_ft);
}

// K time instants (times [K,2]; v_means [K,N,3], v_quats [K,N,4], v_colors [K,N,9]; v_scales [N,3] and v_opac [N] exist
// once): a thread walks the instants of its splat in order -- instant 0 as ACC says, the later ones accumulate (the
// thread reads back its own stores: program order) -- i.e. exactly what K launches in a row did, in one.
template <bool ACC, typename G>
__global__ void __launch_bounds__(256)
prep_bwd_kernel(int Ns, int Nd, int K, const float* __restrict__ times, const long long* __restrict__ d_ncp,
                const float* __restrict__ d_trbf,
                // forward outputs needed for the activation derivatives
                const float* __restrict__ scales, const float* __restrict__ opac,
                // cotangents of the forward outputs (any may be NULL)
                const float* __restrict__ v_means, const float* __restrict__ v_quats,
                const float* __restrict__ v_scales, const float* __restrict__ v_opac,
                const float* __restrict__ v_colors,
                // gradients of the leaves (static)
                float* __restrict__ g_s_xyz, G* __restrict__ g_s_scaling, G* __restrict__ g_s_rotation,
                G* __restrict__ g_s_opacity, G* __restrict__ g_s_fdc, G* __restrict__ g_s_ft,
                // gradients of the leaves (dynamic)
                float* __restrict__ g_d_control, G* __restrict__ g_d_scaling, G* __restrict__ g_d_rotation,
                G* __restrict__ g_d_omega, G* __restrict__ g_d_opacity, G* __restrict__ g_d_fdc,
                G* __restrict__ g_d_ft) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = Ns + Nd;
    if (!ACC) prep_bwd_clear_rows(Ns, Nd, g_d_control);  // (all threads of the wave, before any return)
    if (i >= N) return;
    prep_bwd_instant<ACC, G>(i, Ns, times, d_ncp, d_trbf, scales, opac, v_means, v_quats, v_scales, v_opac, v_colors,
                             g_s_xyz, g_s_scaling, g_s_rotation, g_s_opacity, g_s_fdc, g_s_ft, g_d_control, g_d_scaling,
                             g_d_rotation, g_d_omega, g_d_opacity, g_d_fdc, g_d_ft);
    for (int kk = 1; kk < K; ++kk)
        prep_bwd_instant<true, G>(i, Ns, times + 2 * kk, d_ncp, d_trbf, scales, opac,
                                  v_means ? v_means + (size_t)kk * N * 3 : nullptr,
                                  v_quats ? v_quats + (size_t)kk * N * 4 : nullptr, nullptr, nullptr,
                                  v_colors ? v_colors + (size_t)kk * N * 9 : nullptr, g_s_xyz, g_s_scaling, g_s_rotation,
                                  g_s_opacity, g_s_fdc, g_s_ft, g_d_control, g_d_scaling, g_d_rotation, g_d_omega,
                                  g_d_opacity, g_d_fdc, g_d_ft);
}

}  // namespace mobgs

using namespace mobgs;

template <typename A>
static int prep_fwd_launch(int K, int Ns, int Nd, const float* times, const float* s_xyz, const A* s_scaling,
                           const A* s_rotation, const A* s_opacity, const A* s_fdc, const A* s_ft,
                           const float* d_control, const int64_t* d_ncp, const A* d_scaling, const A* d_rotation,
                           const A* d_omega, const A* d_opacity, const A* d_fdc, const A* d_ft, const float* d_trbf,
                           float* means, float* quats, float* scales, float* opacities, float* colors, void* stream,
                           const char* who) {
    if (Ns < 0 || Nd < 0 || K < 1 || K > 65535) {
        set_error("%s: bad sizes K=%d Ns=%d Nd=%d", who, K, Ns, Nd);
        return MOBGS_E_INVALID;
    }
    const int N = Ns + Nd;
    if (N == 0) return MOBGS_OK;
    const PrepIn<A> in{Ns, Nd, times, s_xyz, s_scaling, s_rotation, s_opacity, s_fdc, s_ft, d_control,
                       (const long long*)d_ncp, d_scaling, d_rotation, d_omega, d_opacity, d_fdc, d_ft, d_trbf};
    hipLaunchKernelGGL(prep_fwd_kernel<A>, dim3((N + 255) / 256, K), dim3(256), 0, (hipStream_t)stream, in, means, quats,
                       scales, opacities, colors);
    return check_launch("prep_fwd_kernel");
}

template <typename G>
static int prep_bwd_launch(int K, int Ns, int Nd, const float* times, const int64_t* d_ncp, const float* d_trbf,
                           const float* scales, const float* opacities, const float* v_means, const float* v_quats,
                           const float* v_scales, const float* v_opacities, const float* v_colors, float* g_s_xyz,
                           G* g_s_scaling, G* g_s_rotation, G* g_s_opacity, G* g_s_fdc, G* g_s_ft, float* g_d_control,
                           G* g_d_scaling, G* g_d_rotation, G* g_d_omega, G* g_d_opacity, G* g_d_fdc, G* g_d_ft,
                           int accumulate, void* stream, const char* who) {
    if (Ns < 0 || Nd < 0 || K < 1) {
        set_error("%s: bad sizes K=%d Ns=%d Nd=%d", who, K, Ns, Nd);
        return MOBGS_E_INVALID;
    }
    const int N = Ns + Nd;
    if (N == 0) return MOBGS_OK;
    if (accumulate)
        hipLaunchKernelGGL((prep_bwd_kernel<true, G>), dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, Ns, Nd, K,
                           times, (const long long*)d_ncp, d_trbf, scales, opacities, v_means, v_quats, v_scales,
                           v_opacities, v_colors, g_s_xyz, g_s_scaling, g_s_rotation, g_s_opacity, g_s_fdc, g_s_ft,
                           g_d_control, g_d_scaling, g_d_rotation, g_d_omega, g_d_opacity, g_d_fdc, g_d_ft);
    else
        hipLaunchKernelGGL((prep_bwd_kernel<false, G>), dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, Ns,
                           Nd, K, times, (const long long*)d_ncp, d_trbf, scales, opacities, v_means, v_quats, v_scales,
                           v_opacities, v_colors, g_s_xyz, g_s_scaling, g_s_rotation, g_s_opacity, g_s_fdc, g_s_ft,
                           g_d_control, g_d_scaling, g_d_rotation, g_d_omega, g_d_opacity, g_d_fdc, g_d_ft);
    return check_launch("prep_bwd_kernel");
}

extern "C" {

int mobgs_prep_fwd_many(int K, int Ns, int Nd, const float* times, const float* s_xyz, const float* s_scaling,
                   const float* s_rotation, const float* s_opacity, const float* s_fdc, const float* s_ft,
                   const float* d_control, const int64_t* d_ncp, const float* d_scaling, const float* d_rotation,
                   const float* d_omega, const float* d_opacity, const float* d_fdc, const float* d_ft,
                   const float* d_trbf, float* means, float* quats, float* scales, float* opacities, float* colors,
                   void* stream) {
    return prep_fwd_launch<float>(K, Ns, Nd, times, s_xyz, s_scaling, s_rotation, s_opacity, s_fdc, s_ft, d_control, d_ncp,
                                  d_scaling, d_rotation, d_omega, d_opacity, d_fdc, d_ft, d_trbf, means, quats, scales,
                                  opacities, colors, stream, "mobgs_prep_fwd");
}

int mobgs_prep_fwd_many_f16(int K, int Ns, int Nd, const float* times, const float* s_xyz, const uint16_t* s_scaling,
                       const uint16_t* s_rotation, const uint16_t* s_opacity, const uint16_t* s_fdc,
                       const uint16_t* s_ft, const float* d_control, const int64_t* d_ncp, const uint16_t* d_scaling,
                       const uint16_t* d_rotation, const uint16_t* d_omega, const uint16_t* d_opacity,
                       const uint16_t* d_fdc, const uint16_t* d_ft, const float* d_trbf, float* means, float* quats,
                       float* scales, float* opacities, float* colors, void* stream) {
    auto H = [](const uint16_t* p) { return reinterpret_cast<const __half*>(p); };
    return prep_fwd_launch<__half>(K, Ns, Nd, times, s_xyz, H(s_scaling), H(s_rotation), H(s_opacity), H(s_fdc), H(s_ft),
                                   d_control, d_ncp, H(d_scaling), H(d_rotation), H(d_omega), H(d_opacity), H(d_fdc),
                                   H(d_ft), d_trbf, means, quats, scales, opacities, colors, stream,
                                   "mobgs_prep_fwd_f16");
}

int mobgs_prep_bwd_many(int K, int Ns, int Nd, const float* times, const int64_t* d_ncp, const float* d_trbf,
                   const float* scales, const float* opacities, const float* v_means, const float* v_quats,
                   const float* v_scales, const float* v_opacities, const float* v_colors, float* g_s_xyz,
                   float* g_s_scaling, float* g_s_rotation, float* g_s_opacity, float* g_s_fdc, float* g_s_ft,
                   float* g_d_control, float* g_d_scaling, float* g_d_rotation, float* g_d_omega, float* g_d_opacity,
                   float* g_d_fdc, float* g_d_ft, int accumulate, void* stream) {
    return prep_bwd_launch<float>(K, Ns, Nd, times, d_ncp, d_trbf, scales, opacities, v_means, v_quats, v_scales,
                                  v_opacities, v_colors, g_s_xyz, g_s_scaling, g_s_rotation, g_s_opacity, g_s_fdc,
                                  g_s_ft, g_d_control, g_d_scaling, g_d_rotation, g_d_omega, g_d_opacity, g_d_fdc,
                                  g_d_ft, accumulate, stream, "mobgs_prep_bwd");
}

int mobgs_prep_bwd_many_f16(int K, int Ns, int Nd, const float* times, const int64_t* d_ncp, const float* d_trbf,
                       const float* scales, const float* opacities, const float* v_means, const float* v_quats,
                       const float* v_scales, const float* v_opacities, const float* v_colors, float* g_s_xyz,
                       uint16_t* g_s_scaling, uint16_t* g_s_rotation, uint16_t* g_s_opacity, uint16_t* g_s_fdc,
                       uint16_t* g_s_ft, float* g_d_control, uint16_t* g_d_scaling, uint16_t* g_d_rotation,
                       uint16_t* g_d_omega, uint16_t* g_d_opacity, uint16_t* g_d_fdc, uint16_t* g_d_ft, int accumulate,
                       void* stream) {
    auto H = [](uint16_t* p) { return reinterpret_cast<__half*>(p); };
    return prep_bwd_launch<__half>(K, Ns, Nd, times, d_ncp, d_trbf, scales, opacities, v_means, v_quats, v_scales,
                                   v_opacities, v_colors, g_s_xyz, H(g_s_scaling), H(g_s_rotation), H(g_s_opacity),
                                   H(g_s_fdc), H(g_s_ft), g_d_control, H(g_d_scaling), H(g_d_rotation), H(g_d_omega),
                                   H(g_d_opacity), H(g_d_fdc), H(g_d_ft), accumulate, stream, "mobgs_prep_bwd_f16");
}


// one time instant (K = 1)
int mobgs_prep_fwd(int Ns, int Nd, const float* times, const float* s_xyz, const float* s_scaling,
                   const float* s_rotation, const float* s_opacity, const float* s_fdc, const float* s_ft,
                   const float* d_control, const int64_t* d_ncp, const float* d_scaling, const float* d_rotation,
                   const float* d_omega, const float* d_opacity, const float* d_fdc, const float* d_ft,
                   const float* d_trbf, float* means, float* quats, float* scales, float* opacities, float* colors,
                   void* stream) {
    return mobgs_prep_fwd_many(1, Ns, Nd, times, s_xyz, s_scaling, s_rotation, s_opacity, s_fdc, s_ft, d_control, d_ncp,
                               d_scaling, d_rotation, d_omega, d_opacity, d_fdc, d_ft, d_trbf, means, quats, scales,
                               opacities, colors, stream);
}

int mobgs_prep_fwd_f16(int Ns, int Nd, const float* times, const float* s_xyz, const uint16_t* s_scaling,
                       const uint16_t* s_rotation, const uint16_t* s_opacity, const uint16_t* s_fdc,
                       const uint16_t* s_ft, const float* d_control, const int64_t* d_ncp, const uint16_t* d_scaling,
                       const uint16_t* d_rotation, const uint16_t* d_omega, const uint16_t* d_opacity,
                       const uint16_t* d_fdc, const uint16_t* d_ft, const float* d_trbf, float* means, float* quats,
                       float* scales, float* opacities, float* colors, void* stream) {
    return mobgs_prep_fwd_many_f16(1, Ns, Nd, times, s_xyz, s_scaling, s_rotation, s_opacity, s_fdc, s_ft, d_control,
                                   d_ncp, d_scaling, d_rotation, d_omega, d_opacity, d_fdc, d_ft, d_trbf, means, quats,
                                   scales, opacities, colors, stream);
}

int mobgs_prep_bwd(int Ns, int Nd, const float* times, const int64_t* d_ncp, const float* d_trbf,
                   const float* scales, const float* opacities, const float* v_means, const float* v_quats,
                   const float* v_scales, const float* v_opacities, const float* v_colors, float* g_s_xyz,
                   float* g_s_scaling, float* g_s_rotation, float* g_s_opacity, float* g_s_fdc, float* g_s_ft,
                   float* g_d_control, float* g_d_scaling, float* g_d_rotation, float* g_d_omega, float* g_d_opacity,
                   float* g_d_fdc, float* g_d_ft, int accumulate, void* stream) {
    return mobgs_prep_bwd_many(1, Ns, Nd, times, d_ncp, d_trbf, scales, opacities, v_means, v_quats, v_scales,
                               v_opacities, v_colors, g_s_xyz, g_s_scaling, g_s_rotation, g_s_opacity, g_s_fdc, g_s_ft,
                               g_d_control, g_d_scaling, g_d_rotation, g_d_omega, g_d_opacity, g_d_fdc, g_d_ft,
                               accumulate, stream);
}

int mobgs_prep_bwd_f16(int Ns, int Nd, const float* times, const int64_t* d_ncp, const float* d_trbf,
                       const float* scales, const float* opacities, const float* v_means, const float* v_quats,
                       const float* v_scales, const float* v_opacities, const float* v_colors, float* g_s_xyz,
                       uint16_t* g_s_scaling, uint16_t* g_s_rotation, uint16_t* g_s_opacity, uint16_t* g_s_fdc,
                       uint16_t* g_s_ft, float* g_d_control, uint16_t* g_d_scaling, uint16_t* g_d_rotation,
                       uint16_t* g_d_omega, uint16_t* g_d_opacity, uint16_t* g_d_fdc, uint16_t* g_d_ft, int accumulate,
                       void* stream) {
    return mobgs_prep_bwd_many_f16(1, Ns, Nd, times, d_ncp, d_trbf, scales, opacities, v_means, v_quats, v_scales,
                                   v_opacities, v_colors, g_s_xyz, g_s_scaling, g_s_rotation, g_s_opacity, g_s_fdc,
                                   g_s_ft, g_d_control, g_d_scaling, g_d_rotation, g_d_omega, g_d_opacity, g_d_fdc,
                                   g_d_ft, accumulate, stream);
}

}  // extern "C"
