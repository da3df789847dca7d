// Host fast path: the bodies of the hot autograd nodes (allocate outputs, gather pointers, ONE C-ABI call) in C++.
//
// At the reference's own operating point (512x288, a few ten thousand splats) a render() step is bound by the host,
// not the device: the kernels of a forward + backward sum to ~0.33 ms while the Python bodies of the eight autograd
// nodes -- ~45 torch.empty calls, ~145 pointer conversions and 17 ctypes calls with 20-40 arguments each -- take
// ~0.45 ms.  This module runs the same bodies natively (at::empty ~0.3 us, no per-argument marshalling).  It launches
// nothing itself: every function ends in the same include/mobgs_hip.h entry point the Python body calls, so the two
// are interchangeable (tests/test_gpu_fastpath.py compares them bit for bit) and the Python bodies remain the
// specification.
//
// No HIP or libmobgs_hip.so symbols at link time: bind() receives the entry points' addresses from the ctypes handle
// (so MOBGS_LIB builds are honoured) and every call receives the raw current stream as an integer.
#include <torch/extension.h>

#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/mobgs_hip.h"

namespace {

using at::Tensor;
using OptT = c10::optional<Tensor>;

struct Api {
    decltype(&mobgs_last_error) last_error = nullptr;
    decltype(&mobgs_record_stride) record_stride = nullptr;
    decltype(&mobgs_prep_fwd_many) prep_fwd = nullptr;
    decltype(&mobgs_prep_fwd_many_f16) prep_fwd_f16 = nullptr;
    decltype(&mobgs_prep_bwd_many) prep_bwd = nullptr;
    decltype(&mobgs_prep_bwd_many_f16) prep_bwd_f16 = nullptr;
    decltype(&mobgs_raster_fwd) raster_fwd = nullptr;
    decltype(&mobgs_raster_fwd_decode) raster_fwd_decode = nullptr;
    decltype(&mobgs_raster_bwd) raster_bwd = nullptr;
    decltype(&mobgs_raster_bwd_decode) raster_bwd_decode = nullptr;
    decltype(&mobgs_raster_bwd_decode_scratch_floats) raster_bwd_decode_scratch_floats = nullptr;
    decltype(&mobgs_raster_bwd_decode_finish) raster_bwd_decode_finish = nullptr;
    decltype(&mobgs_raster_bwd_reduce_decode) raster_bwd_reduce_decode = nullptr;
    decltype(&mobgs_raster_bwd_reduce) raster_bwd_reduce = nullptr;
    decltype(&mobgs_decoder_fwd_channels) decoder_fwd = nullptr;
    decltype(&mobgs_decoder_bwd_channels) decoder_bwd = nullptr;
    decltype(&mobgs_decoder_bwd_blocks) decoder_bwd_blocks = nullptr;
    decltype(&mobgs_project_bwd) project_bwd = nullptr;
    decltype(&mobgs_project_bwd_ex) project_bwd_ex = nullptr;
    decltype(&mobgs_project_bwd_scratch_floats) project_bwd_scratch_floats = nullptr;
    decltype(&mobgs_project_and_bin_speculative) project_and_bin_speculative = nullptr;
    decltype(&mobgs_project_and_bin_fused) project_and_bin_fused = nullptr;
    decltype(&mobgs_prep_project_and_bin_fused) prep_project_and_bin_fused = nullptr;
    decltype(&mobgs_project_prep_bwd_fused) project_prep_bwd_fused = nullptr;
    decltype(&mobgs_fused_seg_keys_len) fused_seg_keys_len = nullptr;
    decltype(&mobgs_tile_order_len) tile_order_len = nullptr;
    decltype(&mobgs_keep_scan_len) keep_scan_len = nullptr;
    decltype(&mobgs_isect_scratch_bytes) isect_scratch_bytes = nullptr;
    decltype(&mobgs_raster_channels_supported) raster_channels_supported = nullptr;
    bool bound = false;
} api;

template <typename F>
void take(const std::unordered_map<std::string, uint64_t>& m, const char* name, F& slot) {
    auto it = m.find(name);
    if (it == m.end() || it->second == 0) throw std::runtime_error(std::string("mobgs fastpath: missing symbol ") + name);
    slot = reinterpret_cast<F>(static_cast<uintptr_t>(it->second));
}

void bind(const std::unordered_map<std::string, uint64_t>& m) {
    // the header this file was compiled against must be the library's (a stale extension passed shifted pointers)
    decltype(&mobgs_abi_version) abi = nullptr;
    take(m, "mobgs_abi_version", abi);
    if (abi() != MOBGS_ABI_VERSION)
        throw std::runtime_error("mobgs fastpath: built against ABI " + std::to_string(MOBGS_ABI_VERSION) +
                                 ", the library reports " + std::to_string(abi()) + " -- rebuild _mobgs_fast.so");
    take(m, "mobgs_last_error", api.last_error);
    take(m, "mobgs_record_stride", api.record_stride);
    take(m, "mobgs_prep_fwd_many", api.prep_fwd);
    take(m, "mobgs_prep_fwd_many_f16", api.prep_fwd_f16);
    take(m, "mobgs_prep_bwd_many", api.prep_bwd);
    take(m, "mobgs_prep_bwd_many_f16", api.prep_bwd_f16);
    take(m, "mobgs_raster_fwd", api.raster_fwd);
    take(m, "mobgs_raster_fwd_decode", api.raster_fwd_decode);
    take(m, "mobgs_raster_bwd", api.raster_bwd);
    take(m, "mobgs_raster_bwd_decode", api.raster_bwd_decode);
    take(m, "mobgs_raster_bwd_decode_scratch_floats", api.raster_bwd_decode_scratch_floats);
    take(m, "mobgs_raster_bwd_decode_finish", api.raster_bwd_decode_finish);
    take(m, "mobgs_raster_bwd_reduce_decode", api.raster_bwd_reduce_decode);
    take(m, "mobgs_raster_bwd_reduce", api.raster_bwd_reduce);
    take(m, "mobgs_decoder_fwd_channels", api.decoder_fwd);
    take(m, "mobgs_decoder_bwd_channels", api.decoder_bwd);
    take(m, "mobgs_decoder_bwd_blocks", api.decoder_bwd_blocks);
    take(m, "mobgs_project_bwd", api.project_bwd);
    take(m, "mobgs_project_bwd_ex", api.project_bwd_ex);
    take(m, "mobgs_project_bwd_scratch_floats", api.project_bwd_scratch_floats);
    take(m, "mobgs_project_and_bin_speculative", api.project_and_bin_speculative);
    take(m, "mobgs_project_and_bin_fused", api.project_and_bin_fused);
    take(m, "mobgs_prep_project_and_bin_fused", api.prep_project_and_bin_fused);
    take(m, "mobgs_project_prep_bwd_fused", api.project_prep_bwd_fused);
    take(m, "mobgs_fused_seg_keys_len", api.fused_seg_keys_len);
    take(m, "mobgs_tile_order_len", api.tile_order_len);
    take(m, "mobgs_keep_scan_len", api.keep_scan_len);
    take(m, "mobgs_isect_scratch_bytes", api.isect_scratch_bytes);
    take(m, "mobgs_raster_channels_supported", api.raster_channels_supported);
    api.bound = true;
}

void check(int rc, const char* what) {
    if (rc != 0) {
        const char* msg = api.last_error ? api.last_error() : "";
        throw std::runtime_error(std::string(what) + " failed (code " + std::to_string(rc) + "): " + msg);
    }
}

// device pointer of a contiguous HIP tensor; the product path has no CPU fallback
inline void* dp(const Tensor& t) {
    if (!t.is_cuda()) throw std::runtime_error("mobgs_amd: tensors must live on a HIP device (device='cuda'); there is no CPU path");
    if (!t.is_contiguous()) throw std::runtime_error("mobgs_amd: internal error, non-contiguous tensor passed to the C ABI");
    return t.data_ptr();
}
inline void* dp(const OptT& t) { return (t.has_value() && t->defined()) ? dp(*t) : nullptr; }
inline const float* fp(const Tensor& t) { return static_cast<const float*>(dp(t)); }
inline const float* fp(const OptT& t) { return static_cast<const float*>(dp(t)); }
inline float* fpw(const Tensor& t) { return static_cast<float*>(dp(t)); }
inline float* fpw(const OptT& t) { return static_cast<float*>(dp(t)); }
inline const int32_t* ip(const Tensor& t) { return static_cast<const int32_t*>(dp(t)); }
inline const int32_t* ip(const OptT& t) { return static_cast<const int32_t*>(dp(t)); }

// float32 + contiguous (no copy when already so)
inline Tensor f32c(const Tensor& t) {
    if (t.scalar_type() == at::kFloat && t.is_contiguous()) return t;
    return t.to(at::kFloat).contiguous();
}
inline OptT f32c(const OptT& t) { return (t.has_value() && t->defined()) ? OptT(f32c(*t)) : OptT(); }

inline void* sp(int64_t stream) { return reinterpret_cast<void*>(static_cast<uintptr_t>(stream)); }
inline const MobgsTuning* tp(int64_t tuning) {
    return reinterpret_cast<const MobgsTuning*>(static_cast<uintptr_t>(tuning));
}

// ---- ops.PrepSplats ------------------------------------------------------------------------------------------------
// attrs: s_scaling, s_rotation, s_opacity, s_fdc, s_ft, d_scaling, d_rotation, d_omega, d_opacity, d_fdc, d_ft
// -> (means, quats, scales, opac, colors, times, d_ncp, d_trbf, n_conversions); the last four are what backward saves
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, int64_t>
prep_fwd(const Tensor& times_in, const Tensor& s_xyz_in, const Tensor& d_control_in, const Tensor& d_ncp_in,
         const Tensor& d_trbf_in, const std::vector<Tensor>& attrs_in, bool half, int64_t stream) {
    if (attrs_in.size() != 11) throw std::runtime_error("prep_fwd: 11 attribute arrays expected");
    const Tensor times = f32c(times_in), s_xyz = f32c(s_xyz_in), d_control = f32c(d_control_in),
                 d_trbf = f32c(d_trbf_in);
    const Tensor d_ncp = (d_ncp_in.scalar_type() == at::kLong && d_ncp_in.is_contiguous())
                             ? d_ncp_in : d_ncp_in.to(at::kLong).contiguous();
    const auto want = half ? at::kHalf : at::kFloat;
    int64_t conversions = 0;
    Tensor a[11];
    for (int i = 0; i < 11; ++i) {
        a[i] = attrs_in[i];
        if (a[i].scalar_type() != want) {
            a[i] = a[i].to(want);
            ++conversions;
        }
        if (!a[i].is_contiguous()) a[i] = a[i].contiguous();
    }
    const int64_t Ns = s_xyz.size(0), Nd = d_control.size(0), N = Ns + Nd;
    const auto opt = times.options().dtype(at::kFloat);
    // times [2]: one instant, outputs [N,*]; times [K,2]: K instants in one launch, means / quats / colors [K,N,*]
    const bool many = times.dim() == 2;
    const int64_t K = many ? times.size(0) : 1;
    if (times.numel() != 2 * K || K < 1) throw std::runtime_error("prep_fwd: times must be [2] or [K,2]");
    Tensor means = many ? at::empty({K, N, 3}, opt) : at::empty({N, 3}, opt),
           quats = many ? at::empty({K, N, 4}, opt) : at::empty({N, 4}, opt), scales = at::empty({N, 3}, opt),
           opac = at::empty({N}, opt), colors = many ? at::empty({K, N, 9}, opt) : at::empty({N, 9}, opt);
    if (half) {
        auto h = [](const Tensor& t) { return static_cast<const uint16_t*>(dp(t)); };
        check(api.prep_fwd_f16((int)K, (int)Ns, (int)Nd, fp(times), fp(s_xyz), h(a[0]), h(a[1]), h(a[2]), h(a[3]), h(a[4]),
                               fp(d_control), static_cast<const int64_t*>(dp(d_ncp)), h(a[5]), h(a[6]), h(a[7]),
                               h(a[8]), h(a[9]), h(a[10]), fp(d_trbf), fpw(means), fpw(quats), fpw(scales),
                               fpw(opac), fpw(colors), sp(stream)),
              "mobgs_prep_fwd");
    } else {
        check(api.prep_fwd((int)K, (int)Ns, (int)Nd, fp(times), fp(s_xyz), fp(a[0]), fp(a[1]), fp(a[2]), fp(a[3]), fp(a[4]),
                           fp(d_control), static_cast<const int64_t*>(dp(d_ncp)), fp(a[5]), fp(a[6]), fp(a[7]),
                           fp(a[8]), fp(a[9]), fp(a[10]), fp(d_trbf), fpw(means), fpw(quats), fpw(scales), fpw(opac),
                           fpw(colors), sp(stream)),
              "mobgs_prep_fwd");
    }
    return {means, quats, scales, opac, colors, times, d_ncp, d_trbf, conversions};
}

// g: the 13 gradient buffers in ops._LEAF_NAMES order (a sink's), or empty -> allocated here (attribute gradients as
// halves when g_half).  Returns the 13 buffers.
std::vector<Tensor> prep_bwd(int64_t Ns, int64_t Nd, const Tensor& times, const Tensor& d_ncp, const Tensor& d_trbf,
                             const Tensor& scales, const Tensor& opac, const OptT& v_means, const OptT& v_quats,
                             const OptT& v_scales, const OptT& v_opac, const OptT& v_colors, std::vector<Tensor> g,
                             bool g_half, int64_t accumulate, int64_t stream) {
    if (g.empty()) {
        const auto f = times.options().dtype(at::kFloat);
        const auto a = times.options().dtype(g_half ? at::kHalf : at::kFloat);
        g = {at::empty({Ns, 3}, f), at::empty({Ns, 3}, a), at::empty({Ns, 4}, a), at::empty({Ns, 1}, a),
             at::empty({Ns, 6}, a), at::empty({Ns, 3}, a), at::empty({Nd, 12, 3}, f), at::empty({Nd, 3}, a),
             at::empty({Nd, 4}, a), at::empty({Nd, 4}, a), at::empty({Nd, 1}, a), at::empty({Nd, 6}, a),
             at::empty({Nd, 3}, a)};
    } else if (g.size() != 13) {
        throw std::runtime_error("prep_bwd: 13 gradient buffers expected");
    }
    const OptT c0 = f32c(v_means), c1 = f32c(v_quats), c2 = f32c(v_scales), c3 = f32c(v_opac), c4 = f32c(v_colors);
    const int64_t K = times.dim() == 2 ? times.size(0) : 1;
    if (g_half) {
        auto h = [](const Tensor& t) { return static_cast<uint16_t*>(dp(t)); };
        check(api.prep_bwd_f16((int)K, (int)Ns, (int)Nd, fp(times), static_cast<const int64_t*>(dp(d_ncp)), fp(d_trbf),
                               fp(scales), fp(opac), fp(c0), fp(c1), fp(c2), fp(c3), fp(c4), fpw(g[0]), h(g[1]),
                               h(g[2]), h(g[3]), h(g[4]), h(g[5]), fpw(g[6]), h(g[7]), h(g[8]), h(g[9]), h(g[10]),
                               h(g[11]), h(g[12]), (int)accumulate, sp(stream)),
              "mobgs_prep_bwd");
    } else {
        check(api.prep_bwd((int)K, (int)Ns, (int)Nd, fp(times), static_cast<const int64_t*>(dp(d_ncp)), fp(d_trbf),
                           fp(scales), fp(opac), fp(c0), fp(c1), fp(c2), fp(c3), fp(c4), fpw(g[0]), fpw(g[1]),
                           fpw(g[2]), fpw(g[3]), fpw(g[4]), fpw(g[5]), fpw(g[6]), fpw(g[7]), fpw(g[8]), fpw(g[9]),
                           fpw(g[10]), fpw(g[11]), fpw(g[12]), (int)accumulate, sp(stream)),
              "mobgs_prep_bwd");
    }
    return g;
}

// ---- rendering._Rasterize ------------------------------------------------------------------------------------------
// One compositing launch.  records / reach: pass the tensors to (re)use, or None to have them allocated.
// dec_intr / dec_c2w / dec_w1 / dec_w2 (all or none): the Sandwich decoder runs as the kernel's epilogue
// (mobgs_raster_fwd_decode) and rgb [C,3,H,W] / depth [C,H,W] come back too.
// -> (records, render [C,H,W,D], alphas [C,H,W], last_ids, reach, rgb | None, depth | None)
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, OptT, OptT>
raster_fwd(int64_t C, int64_t N, int64_t channels, int64_t width, int64_t height, const Tensor& means2d,
           const Tensor& conics, const OptT& colors, int64_t colors_per_camera, const Tensor& opacities,
           int64_t opac_per_camera, const OptT& extra, const OptT& bg, const Tensor& radii, const Tensor& tile_offsets,
           const OptT& tile_order, const Tensor& flatten_arena, const OptT& records_in, const OptT& reach_in,
           int64_t tuning, int64_t stream, const OptT& dec_intr, const OptT& dec_c2w, const OptT& dec_w1,
           const OptT& dec_w2) {
    const bool has_extra = extra.has_value() && extra->defined();
    const int64_t D = channels + (has_extra ? 1 : 0);
    const auto f = means2d.options().dtype(at::kFloat);
    Tensor records = (records_in.has_value() && records_in->defined())
                         ? *records_in : at::empty({C * N, (int64_t)api.record_stride((int)D)}, f);
    const int64_t arena = std::max<int64_t>(flatten_arena.numel(), 1);
    Tensor reach = (reach_in.has_value() && reach_in->defined() && reach_in->numel() >= flatten_arena.numel())
                       ? *reach_in : at::empty({arena}, f.dtype(at::kByte));
    Tensor render = at::empty({C, height, width, D}, f), alphas = at::empty({C, height, width}, f),
           last_ids = at::empty({C, height, width}, f.dtype(at::kInt));
    if (dec_intr.has_value() && dec_intr->defined()) {
        Tensor rgb = at::empty({C, 3, height, width}, f), depth = at::empty({C, height, width}, f);
        const int intr_stride = (C > 1 && dec_intr->numel() == 4 * C) ? 4 : 0;
        const int c2w_stride = (C > 1 && dec_c2w->dim() == 3) ? (int)(dec_c2w->numel() / C) : 0;
        check(api.raster_fwd_decode((int)C, (int)N, (int)channels, (int)width, (int)height, fp(means2d), fp(conics),
                                    fp(colors), (int)colors_per_camera, fp(opacities), (int)opac_per_camera, fp(extra),
                                    fp(bg), ip(radii), ip(tile_offsets), ip(tile_order), ip(flatten_arena), fpw(records),
                                    fpw(render), fpw(alphas), static_cast<int32_t*>(dp(last_ids)),
                                    static_cast<uint8_t*>(dp(reach)), fp(dec_intr), intr_stride, fp(dec_c2w), c2w_stride,
                                    fp(dec_w1), fp(dec_w2), fpw(rgb), fpw(depth), tp(tuning), sp(stream)),
              "mobgs_raster_fwd_decode");
        return {records, render, alphas, last_ids, reach, OptT(rgb), OptT(depth)};
    }
    check(api.raster_fwd((int)C, (int)N, (int)channels, (int)width, (int)height, fp(means2d), fp(conics), fp(colors),
                         (int)colors_per_camera, fp(opacities), (int)opac_per_camera, fp(extra), fp(bg), ip(radii),
                         ip(tile_offsets), ip(tile_order), ip(flatten_arena), fpw(records), fpw(render), fpw(alphas),
                         static_cast<int32_t*>(dp(last_ids)), static_cast<uint8_t*>(dp(reach)), tp(tuning),
                         sp(stream)),
          "mobgs_raster_fwd");
    return {records, render, alphas, last_ids, reach, OptT(), OptT()};
}

// -> zero-filled gradient slots [max(n_isects,1) + 1, stride] with the per-entry records written by the kernel; the
// first word of the extra last row is the any_record flag of include/mobgs_hip.h (zeroed by the same fill)
Tensor raster_bwd(int64_t C, int64_t N, int64_t channels, int64_t has_extra, int64_t width, int64_t height,
                  int64_t n_isects, const Tensor& records, const OptT& bg, const Tensor& radii, const Tensor& means2d,
                  const Tensor& cum_tiles, const Tensor& keep_scan, const Tensor& tile_offsets, const OptT& tile_order,
                  const Tensor& flatten_ids, const Tensor& alphas, const Tensor& last_ids, const Tensor& v_render_in,
                  const OptT& v_alphas_in, const OptT& reach, int64_t tuning, int64_t stream, bool cover) {
    const Tensor v_render = f32c(v_render_in);
    const OptT v_alphas = f32c(v_alphas_in);
    const int64_t rows = std::max<int64_t>(n_isects, 1), stride = records.size(1);
    // cover (MobgsTuning.cover_slots, set by the caller in `tuning`): the kernel writes every slot -- no fill, no flag
    Tensor slots = cover ? at::empty({rows + 1, stride}, records.options()) : at::zeros({rows + 1, stride}, records.options());
    int32_t* flag = cover ? nullptr : reinterpret_cast<int32_t*>(fpw(slots) + rows * stride);
    check(api.raster_bwd((int)C, (int)N, (int)channels, (int)has_extra, (int)width, (int)height, fp(records), fp(bg),
                         ip(radii), fp(means2d), ip(cum_tiles), ip(keep_scan), ip(tile_offsets), ip(tile_order),
                         ip(flatten_ids), fp(alphas), ip(last_ids), fp(v_render), fp(v_alphas), fpw(slots),
                         static_cast<const uint8_t*>(dp(reach)), flag, tp(tuning), sp(stream)),
          "mobgs_raster_bwd");
    return slots;
}

// The same with the decoder's backward pass as the kernel's prologue (mobgs_raster_bwd_decode): -> (slots, partial rows)
std::tuple<Tensor, Tensor>
raster_bwd_decode(int64_t C, int64_t N, int64_t width, int64_t height, int64_t n_isects, const Tensor& records,
                  const OptT& bg, const Tensor& radii, const Tensor& cum_tiles, const Tensor& keep_scan,
                  const Tensor& tile_offsets, const OptT& tile_order, const Tensor& flatten_ids, const Tensor& render,
                  const Tensor& alphas, const Tensor& last_ids, const OptT& v_rgb_in, const OptT& v_depth_in,
                  const OptT& v_alphas_in, const Tensor& intr, const Tensor& c2w, const Tensor& w1, const Tensor& w2,
                  const OptT& reach, int64_t tuning, int64_t stream, bool cover) {
    const auto f = records.options();
    const Tensor v_rgb = (v_rgb_in.has_value() && v_rgb_in->defined()) ? f32c(*v_rgb_in) : at::zeros({C, 3, height, width}, f);
    const OptT v_depth = f32c(v_depth_in);
    const OptT v_alphas = f32c(v_alphas_in);
    const int64_t rows = std::max<int64_t>(n_isects, 1), stride = records.size(1);
    Tensor slots = cover ? at::empty({rows + 1, stride}, f) : at::zeros({rows + 1, stride}, f);
    int32_t* flag = cover ? nullptr : reinterpret_cast<int32_t*>(fpw(slots) + rows * stride);
    Tensor partial = at::empty({(int64_t)api.raster_bwd_decode_scratch_floats((int)C, (int)width, (int)height)}, f);
    const int64_t intr_stride = (C > 1 && intr.numel() == 4 * C) ? 4 : 0;
    const int64_t c2w_stride = (C > 1 && c2w.dim() == 3) ? c2w.numel() / C : 0;
    check(api.raster_bwd_decode((int)C, (int)N, (int)width, (int)height, fp(records), fp(bg), ip(radii), ip(cum_tiles),
                                ip(keep_scan), ip(tile_offsets), ip(tile_order), ip(flatten_ids), fp(render), fp(alphas),
                                ip(last_ids), fp(v_rgb), fp(v_depth), fp(v_alphas), fp(intr), (int)intr_stride, fp(c2w),
                                (int)c2w_stride, fp(w1), fp(w2), fpw(slots), static_cast<const uint8_t*>(dp(reach)), flag,
                                fpw(partial), tp(tuning), sp(stream)),
          "mobgs_raster_bwd_decode");
    return {slots, partial};
}

// ... and the fixed-order sum of its partial rows (mobgs_raster_bwd_decode_finish): -> (g_c2w | None, g_w1, g_w2);
// g_w1 / g_w2: a sink's buffers (accumulate as given) or None -> allocated here and overwritten.
std::tuple<OptT, Tensor, Tensor>
raster_bwd_decode_finish(int64_t C, int64_t width, int64_t height, const Tensor& partial, const Tensor& c2w, const Tensor& w1,
                         const Tensor& w2, bool c2w_needs_grad, const OptT& g_w1_in, const OptT& g_w2_in, int64_t accumulate,
                         int64_t stream) {
    const bool sunk = g_w1_in.has_value() && g_w1_in->defined();
    Tensor g_w1 = sunk ? *g_w1_in : at::empty_like(w1);
    Tensor g_w2 = sunk ? *g_w2_in : at::empty_like(w2);
    OptT g_c2w = c2w_needs_grad ? OptT(at::empty_like(c2w)) : OptT();
    const int64_t c2w_stride = (C > 1 && c2w.dim() == 3) ? c2w.numel() / C : 0;
    check(api.raster_bwd_decode_finish((int)C, (int)width, (int)height, fpw(partial), (int)c2w_stride, fpw(g_w1), fpw(g_w2),
                                       fpw(g_c2w), g_c2w.has_value() ? (int)(g_c2w->numel() / (c2w_stride ? C : 1)) : 0,
                                       sunk ? (int)accumulate : 0, sp(stream)),
          "mobgs_raster_bwd_decode_finish");
    return {g_c2w, g_w1, g_w2};
}

// raster_bwd_reduce + raster_bwd_decode_finish in one launch (mobgs_raster_bwd_reduce_decode):
// -> (v_means2d, v_conics, v_opac, v_colors, v_extra, g_c2w | None, g_w1, g_w2)
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, OptT, Tensor, Tensor>
raster_bwd_reduce_decode(int64_t C, int64_t N, int64_t width, int64_t height, const Tensor& records, const Tensor& cum_tiles,
                         const Tensor& keep_scan, const Tensor& slots, const OptT& tiles_per_gauss, int64_t flag_ptr,
                         const Tensor& partial, const Tensor& c2w, const Tensor& w1, const Tensor& w2, bool c2w_needs_grad,
                         const OptT& g_w1_in, const OptT& g_w2_in, int64_t accumulate, int64_t stream) {
    const auto f = slots.options();
    Tensor v_means2d = at::empty({C, N, 2}, f), v_conics = at::empty({C, N, 3}, f), v_opac = at::empty({C, N}, f),
           v_colors = at::empty({C, N, 9}, f), v_extra = at::empty({C, N}, f);
    // flag_ptr: -1 = the first word of the slots' extra row (stage 1 set it), else the address of a device int32 that is 0 when
    // no slot may be read (the lists' total: cover_slots mode) or 0 = none
    const int32_t* flag = flag_ptr == -1 ? reinterpret_cast<const int32_t*>(fp(slots) + (slots.size(0) - 1) * slots.size(1))
                                         : reinterpret_cast<const int32_t*>(static_cast<uintptr_t>(flag_ptr));
    const bool sunk = g_w1_in.has_value() && g_w1_in->defined();
    Tensor g_w1 = sunk ? *g_w1_in : at::empty_like(w1);
    Tensor g_w2 = sunk ? *g_w2_in : at::empty_like(w2);
    OptT g_c2w = c2w_needs_grad ? OptT(at::empty_like(c2w)) : OptT();
    const int64_t c2w_stride = (C > 1 && c2w.dim() == 3) ? c2w.numel() / C : 0;
    check(api.raster_bwd_reduce_decode((int)C, (int)N, fp(records), ip(cum_tiles), ip(keep_scan), fp(slots), flag,
                                       fpw(v_means2d), fpw(v_conics), fpw(v_opac), fpw(v_colors), fpw(v_extra),
                                       ip(tiles_per_gauss), (int)width, (int)height, fp(partial), (int)c2w_stride, fpw(g_w1),
                                       fpw(g_w2), fpw(g_c2w),
                                       g_c2w.has_value() ? (int)(g_c2w->numel() / (c2w_stride ? C : 1)) : 0,
                                       sunk ? (int)accumulate : 0, sp(stream)),
          "mobgs_raster_bwd_reduce_decode");
    return {v_means2d, v_conics, v_opac, v_colors, v_extra, g_c2w, g_w1, g_w2};
}

// -> (v_means2d [C,N,2], v_conics [C,N,3], v_opac [C,N], v_colors [C,N,channels], v_extra [C,N] | None)
std::tuple<Tensor, Tensor, Tensor, Tensor, OptT>
raster_bwd_reduce(int64_t C, int64_t N, int64_t channels, int64_t has_extra, const Tensor& records,
                  const Tensor& cum_tiles, const Tensor& keep_scan, const Tensor& slots, int64_t stream,
                  const OptT& tiles_per_gauss, int64_t flag_ptr) {
    const auto f = slots.options();
    Tensor v_means2d = at::empty({C, N, 2}, f), v_conics = at::empty({C, N, 3}, f), v_opac = at::empty({C, N}, f),
           v_colors = at::empty({C, N, channels}, f);
    OptT v_extra = has_extra ? OptT(at::empty({C, N}, f)) : OptT();
    const int32_t* flag = flag_ptr == -1 ? reinterpret_cast<const int32_t*>(fp(slots) + (slots.size(0) - 1) * slots.size(1))
                                         : reinterpret_cast<const int32_t*>(static_cast<uintptr_t>(flag_ptr));   // (as above)
    check(api.raster_bwd_reduce((int)C, (int)N, (int)channels, (int)has_extra, fp(records), ip(cum_tiles), ip(keep_scan),
                                fp(slots), flag, fpw(v_means2d), fpw(v_conics), fpw(v_opac), fpw(v_colors), fpw(v_extra),
                                ip(tiles_per_gauss), sp(stream)),
          "mobgs_raster_bwd_reduce");
    return {v_means2d, v_conics, v_opac, v_colors, v_extra};
}

// ---- ops.Decode ----------------------------------------------------------------------------------------------------
// feat_hw [H,W,CF] -> (rgb [3,H,W], depth [H,W] | None); a batch feat_hw [C,H,W,CF] -> ([C,3,H,W], [C,H,W] | None) in one
// launch: ray map [6,H,W] shared or [C,6,H,W]; intrinsics [4] or [C,4]; pose [3|4,4] shared or [C,3|4,4]
struct DecStrides {
    int64_t C, rays, intr, c2w;
};
static DecStrides decoder_strides(int64_t H, int64_t W, int64_t CF, const Tensor& feat_hw, const OptT& rays,
                                  const OptT& intr, const OptT& c2w) {
    DecStrides d{feat_hw.numel() / (H * W * CF), 0, 0, 0};
    if (d.C > 1) {
        if (rays.has_value() && rays->defined() && rays->numel() == d.C * 6 * H * W) d.rays = 6 * H * W;
        if (intr.has_value() && intr->defined() && intr->numel() == d.C * 4) d.intr = 4;
        if (c2w.has_value() && c2w->defined() && c2w->dim() == 3 && c2w->size(0) == d.C) d.c2w = c2w->numel() / d.C;
    }
    return d;
}

// chan_n > 0: channels [chan_c0, chan_c0 + chan_n) of the image come back as a contiguous tensor [..., chan_n] of their own
std::tuple<Tensor, OptT, OptT> decoder_fwd(int64_t H, int64_t W, int64_t CF, bool has_depth, const Tensor& feat_hw,
                                           const OptT& alphas, const OptT& rays, const OptT& intr, const OptT& c2w,
                                           const Tensor& w1, const Tensor& w2, int64_t stream, int64_t chan_c0,
                                           int64_t chan_n) {
    const auto f = feat_hw.options();
    const DecStrides d = decoder_strides(H, W, CF, feat_hw, rays, intr, c2w);
    const bool batch = feat_hw.dim() == 4;
    Tensor rgb = batch ? at::empty({d.C, 3, H, W}, f) : at::empty({3, H, W}, f);
    OptT depth = has_depth ? OptT(batch ? at::empty({d.C, H, W}, f) : at::empty({H, W}, f)) : OptT();
    OptT chan;
    if (chan_n > 0) {
        std::vector<int64_t> shape(feat_hw.sizes().begin(), feat_hw.sizes().end());
        shape.back() = chan_n;
        chan = at::empty(shape, f);
    }
    check(api.decoder_fwd((int)d.C, (int)(H * W), (int)CF, has_depth ? 1 : 0, (int)W, fp(feat_hw), fp(alphas), fp(rays),
                          d.rays, fp(intr), (int)d.intr, fp(c2w), (int)d.c2w, fp(w1), fp(w2), fpw(rgb), fpw(depth),
                          fpw(chan), (int)chan_c0, (int)chan_n, sp(stream)),
          "mobgs_decoder_fwd");
    return {rgb, depth, chan};
}

// g_w1 / g_w2: a sink's buffers (accumulate as given) or None -> allocated here and overwritten.
// -> (v_feat, v_alphas | None, v_rays | None, g_c2w | None, g_w1, g_w2)
std::tuple<Tensor, OptT, OptT, OptT, Tensor, Tensor>
decoder_bwd(int64_t H, int64_t W, int64_t CF, bool has_depth, const Tensor& feat_hw, const OptT& alphas,
            const OptT& rays, const OptT& intr, const OptT& c2w, const Tensor& w1, const Tensor& w2,
            const OptT& v_rgb_in, const OptT& v_depth_in, std::vector<int64_t> feat_shape, bool rays_need_grad,
            bool c2w_needs_grad, const OptT& g_w1_in, const OptT& g_w2_in, int64_t accumulate, int64_t stream,
            const OptT& v_chan_in, int64_t chan_c0) {
    // v_chan (optional): the cotangent of the channels decoder_fwd handed out; written into v_feat by the kernel
    const OptT v_chan = f32c(v_chan_in);
    const int64_t chan_n = (v_chan.has_value() && v_chan->defined()) ? v_chan->size(-1) : 0;
    const auto f = feat_hw.options();
    const int64_t P = H * W;
    const DecStrides d = decoder_strides(H, W, CF, feat_hw, rays, intr, c2w);
    Tensor v_rgb = (v_rgb_in.has_value() && v_rgb_in->defined()) ? f32c(*v_rgb_in) : at::zeros({d.C, 3, H, W}, f);
    OptT v_depth = has_depth ? f32c(v_depth_in) : OptT();
    Tensor v_feat = at::empty(feat_shape, f);
    OptT v_alphas = has_depth ? OptT(at::empty(alphas->sizes(), f)) : OptT();
    OptT v_rays = rays_need_grad ? OptT(at::empty_like(*rays)) : OptT();
    OptT g_c2w = c2w_needs_grad ? OptT(at::empty_like(*c2w)) : OptT();
    Tensor partial = at::empty({d.C * (int64_t)api.decoder_bwd_blocks((int)P), 102}, f);
    const bool sunk = g_w1_in.has_value() && g_w1_in->defined();
    Tensor g_w1 = sunk ? *g_w1_in : at::empty_like(w1);
    Tensor g_w2 = sunk ? *g_w2_in : at::empty_like(w2);
    check(api.decoder_bwd((int)d.C, (int)P, (int)CF, has_depth ? 1 : 0, (int)W, fp(feat_hw), fp(alphas), fp(rays), d.rays,
                          fp(intr), (int)d.intr, fp(c2w), (int)d.c2w, fp(w1), fp(w2), fp(v_rgb), fp(v_depth),
                          fpw(v_feat), fpw(v_alphas), fpw(v_rays), fpw(partial), fpw(g_w1), fpw(g_w2), fpw(g_c2w),
                          g_c2w.has_value() ? (int)(g_c2w->numel() / (d.c2w ? d.C : 1)) : 0,
                          sunk ? (int)accumulate : 0, fp(v_chan), (int)chan_c0, (int)chan_n, sp(stream)),
          "mobgs_decoder_bwd");
    return {v_feat, v_alphas, v_rays, g_c2w, g_w1, g_w2};
}

// ---- rendering._Project.backward -----------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor>
project_bwd(int64_t width, int64_t height, double eps2d, const Tensor& means, const Tensor& quats, const Tensor& scales,
            const Tensor& viewmats, const Tensor& Ks, const Tensor& radii, const Tensor& conics, const OptT& v_means2d,
            const OptT& v_depths, const OptT& v_conics, int64_t stream) {
    // means [N,3] / quats [N,4] shared by the cameras, or [C,N,3] / [C,N,4] (MobgsTuning.geometry_per_camera)
    const int64_t C = viewmats.size(0), N = means.size(-2);
    const int per_cam = means.dim() == 3 ? 1 : 0;
    Tensor v_means = at::empty_like(means), v_quats = at::empty_like(quats), v_scales = at::empty_like(scales),
           v_viewmats = at::empty_like(viewmats);
    Tensor partial = at::empty({(int64_t)api.project_bwd_scratch_floats((int)C, (int)N)}, means.options());
    const OptT g2 = f32c(v_means2d), gd = f32c(v_depths), gc = f32c(v_conics);
    check(api.project_bwd_ex((int)C, (int)N, per_cam, fp(means), fp(quats), fp(scales), fp(viewmats), fp(Ks), (int)width,
                             (int)height, (float)eps2d, ip(radii), fp(conics), fp(g2), fp(gd), fp(gc), fpw(v_means),
                             fpw(v_quats), fpw(v_scales), fpw(v_viewmats), fpw(partial), sp(stream)),
          "mobgs_project_bwd");
    return {v_means, v_quats, v_scales, v_viewmats};
}

// the five state buffers of rendering._PrepProjectAndBin.forward in one call (means, quats, scales, opac, colour token)
std::vector<Tensor> prep_state_buffers(int64_t N, const Tensor& like) {
    const auto f = like.options().dtype(at::kFloat);
    return {at::empty({N, 3}, f), at::empty({N, 4}, f), at::empty({N, 3}, f), at::empty({N}, f), at::empty({N, 9}, f)};
}

// ---- rendering._PrepProjectAndBin.backward: projection backward + prep backward in one launch -----------------------------
// g: a sink's 13 float32 gradient buffers (accumulate = 1) or empty -> allocated here.  -> (the 13 buffers, v_viewmats)
std::tuple<std::vector<Tensor>, OptT>
project_prep_bwd(int64_t width, int64_t height, double eps2d, const Tensor& means, const Tensor& quats,
                 const Tensor& scales, const Tensor& viewmats, const Tensor& Ks, const Tensor& radii,
                 const Tensor& conics, const OptT& v_means2d, const OptT& v_depths, const OptT& v_conics,
                 const OptT& x_means, const OptT& x_quats, const OptT& x_scales, int64_t Ns, int64_t Nd,
                 const Tensor& times, const Tensor& d_ncp, const Tensor& d_trbf, const Tensor& opac, const OptT& v_opac,
                 const OptT& v_colors, std::vector<Tensor> g, int64_t accumulate, int64_t stream, bool want_viewmats) {
    const int64_t N = means.size(0);
    const auto f = means.options().dtype(at::kFloat);
    if (g.empty()) {
        g = {at::empty({Ns, 3}, f), at::empty({Ns, 3}, f), at::empty({Ns, 4}, f), at::empty({Ns, 1}, f),
             at::empty({Ns, 6}, f), at::empty({Ns, 3}, f), at::empty({Nd, 12, 3}, f), at::empty({Nd, 3}, f),
             at::empty({Nd, 4}, f), at::empty({Nd, 4}, f), at::empty({Nd, 1}, f), at::empty({Nd, 6}, f),
             at::empty({Nd, 3}, f)};
        accumulate = 0;
    } else if (g.size() != 13) {
        throw std::runtime_error("project_prep_bwd: 13 gradient buffers expected");
    }
    OptT v_viewmats, partial;
    if (want_viewmats) {
        v_viewmats = at::empty_like(viewmats);
        partial = at::empty({(int64_t)api.project_bwd_scratch_floats(1, (int)N)}, f);
    }
    const OptT g2 = f32c(v_means2d), gd = f32c(v_depths), gc = f32c(v_conics), xm = f32c(x_means), xq = f32c(x_quats),
               xs = f32c(x_scales), vo = f32c(v_opac), vc = f32c(v_colors);
    MobgsLeafGrads lg{fpw(g[0]), fpw(g[1]), fpw(g[2]), fpw(g[3]), fpw(g[4]), fpw(g[5]), fpw(g[6]),
                      fpw(g[7]), fpw(g[8]), fpw(g[9]), fpw(g[10]), fpw(g[11]), fpw(g[12])};
    check(api.project_prep_bwd_fused((int)N, fp(means), fp(quats), fp(scales), fp(viewmats), fp(Ks), (int)width,
                                     (int)height, (float)eps2d, ip(radii), fp(conics), fp(g2), fp(gd), fp(gc), fp(xm),
                                     fp(xq), fp(xs), fpw(v_viewmats), fpw(partial), (int)Ns, (int)Nd, fp(times),
                                     static_cast<const int64_t*>(dp(d_ncp)), fp(d_trbf), fp(opac), fp(vo), fp(vc), &lg,
                                     (int)accumulate, sp(stream)),
          "mobgs_project_prep_bwd_fused");
    return {g, v_viewmats};
}

// ---- rendering._ProjectAndBin.forward, speculative binning ---------------------------------------------------------
// Allocates every output / arena and makes the ONE orchestrator call.  -> (rc, [radii, means2d, depths, conics,
// tiles_per_gauss, cum_tiles, tile_offsets, keep_scan, flatten_ids], tile_order | None, isect_ids | None,
// records | None).  rc: 0 = counts arrive through the polled pinned row, 1 = by asynchronous copy (record an event).
std::tuple<int64_t, std::vector<Tensor>, OptT, OptT, OptT>
project_and_bin_speculative(const Tensor& means, const Tensor& quats, const Tensor& scales, const Tensor& viewmats,
                            const Tensor& Ks, const Tensor& opac, int64_t width, int64_t height, double eps2d,
                            double near_plane, double far_plane, double radius_clip, int64_t cull,
                            bool want_isect_ids, bool tile_schedule, const OptT& pack_colors, int64_t cap_box,
                            int64_t cap_listed, int64_t len_hint, int64_t stats_row, int64_t seq, int64_t tuning,
                            int64_t stream, int64_t seg_stride, const OptT& enum_order,
                            const std::vector<Tensor>& prep) {
    // prep (empty, or the 16 float32 / int64 contiguous inputs of ops.PrepSplats: times, s_xyz, s_scaling, s_rotation,
    // s_opacity, s_fdc, s_ft, d_control, d_ncp, d_scaling, d_rotation, d_omega, d_opacity, d_fdc, d_ft, d_trbf): the
    // projection kernel builds the per-splat state itself -- means / quats / scales / opac are then OUTPUT buffers and
    // the packed records carry the colour features (mobgs_prep_project_and_bin_fused)
    const bool fused_prep = !prep.empty();
    if (fused_prep && prep.size() != 16) throw std::runtime_error("project_and_bin: 16 prep inputs expected");
    const int64_t C = viewmats.size(0), N = means.size(-2);  // means [N,3] or, with geometry_per_camera, [C,N,3]
    const int64_t tile_w = (width + 15) / 16, tile_h = (height + 15) / 16, nt = C * tile_w * tile_h;
    const auto f = means.options().dtype(at::kFloat);
    const auto i32 = f.dtype(at::kInt);
    const auto i64 = f.dtype(at::kLong);
    Tensor radii = at::empty({C, N}, i32), means2d = at::empty({C, N, 2}, f), depths = at::empty({C, N}, f),
           conics = at::empty({C, N, 3}, f), tiles_per_gauss = at::empty({C, N}, i32),
           cum_tiles = at::empty({C * N + 1}, i32), tile_offsets = at::empty({nt + 1}, i32),
           stats_dev = at::empty({3}, i64);
    OptT tile_order = tile_schedule ? OptT(at::empty({(int64_t)api.tile_order_len((int)nt)}, i32)) : OptT();
    OptT records;
    int64_t pack_ch = 0;
    if (fused_prep) {
        records = at::empty({C * N, (int64_t)api.record_stride(10)}, f);
    } else if (pack_colors.has_value() && pack_colors->defined()) {
        pack_ch = pack_colors->size(-1);
        if (api.raster_channels_supported((int)pack_ch + 1))
            records = at::empty({C * N, (int64_t)api.record_stride((int)pack_ch + 1)}, f);
    }
    Tensor keep_scan = at::empty({(int64_t)api.keep_scan_len((int)cap_box)}, i32);
    Tensor scratch = at::empty({(int64_t)api.isect_scratch_bytes((int)(C * N), (int)nt, (int)cap_box)},
                               f.dtype(at::kByte));
    // seg_stride > 0: single-pass lists (mobgs_project_and_bin_fused) -- the key arena is [tile][8][seg_stride]
    Tensor flatten_ids = at::empty({cap_listed}, i32),
           sort_keys = at::empty({seg_stride > 0 ? (int64_t)api.fused_seg_keys_len((int)nt, (int)seg_stride) : cap_listed}, i64);
    OptT isect_ids = want_isect_ids ? OptT(at::empty({cap_listed}, i64)) : OptT();
    const bool pack = records.has_value();
    int rc;
    if (fused_prep) {
        MobgsPrepInputs pi;
        pi.Ns = (int32_t)prep[1].size(0);
        pi.Nd = (int32_t)prep[7].size(0);
        pi.times = fp(prep[0]);
        pi.s_xyz = fp(prep[1]); pi.s_scaling = fp(prep[2]); pi.s_rotation = fp(prep[3]); pi.s_opacity = fp(prep[4]);
        pi.s_fdc = fp(prep[5]); pi.s_ft = fp(prep[6]);
        pi.d_control = fp(prep[7]);
        pi.d_ncp = static_cast<const int64_t*>(dp(prep[8]));
        pi.d_scaling = fp(prep[9]); pi.d_rotation = fp(prep[10]); pi.d_omega = fp(prep[11]); pi.d_opacity = fp(prep[12]);
        pi.d_fdc = fp(prep[13]); pi.d_ft = fp(prep[14]); pi.d_trbf = fp(prep[15]);
        if (C != 1 || pi.Ns + pi.Nd != N) throw std::runtime_error("project_and_bin: fused prep needs one camera and Ns + Nd = N");
        rc = api.prep_project_and_bin_fused(
            &pi, fpw(means), fpw(quats), fpw(scales), fp(viewmats), fp(Ks), fpw(opac), (int)width, (int)height,
            (float)eps2d, (float)near_plane, (float)far_plane, (float)radius_clip, (int)cull,
            static_cast<int32_t*>(dp(radii)), fpw(means2d), fpw(depths), fpw(conics),
            static_cast<int32_t*>(dp(tiles_per_gauss)), static_cast<int32_t*>(dp(cum_tiles)),
            static_cast<int32_t*>(dp(tile_offsets)), static_cast<int32_t*>(dp(tile_order)),
            static_cast<int64_t*>(dp(stats_dev)), (int)cap_box, static_cast<int32_t*>(dp(keep_scan)), dp(scratch),
            cap_listed, static_cast<int32_t*>(dp(flatten_ids)), static_cast<uint64_t*>(dp(sort_keys)), (int)seg_stride,
            ip(enum_order), static_cast<uint64_t*>(dp(isect_ids)), len_hint,
            reinterpret_cast<int64_t*>(static_cast<uintptr_t>(stats_row)), seq, fpw(records), tp(tuning), sp(stream));
    } else if (seg_stride > 0)
        rc = api.project_and_bin_fused(
            (int)C, (int)N, fp(means), fp(quats), fp(scales), fp(viewmats), fp(Ks), fp(opac), opac.dim() == 2 ? 1 : 0,
            (int)width, (int)height, (float)eps2d, (float)near_plane, (float)far_plane, (float)radius_clip, (int)cull,
            static_cast<int32_t*>(dp(radii)), fpw(means2d), fpw(depths), fpw(conics),
            static_cast<int32_t*>(dp(tiles_per_gauss)), static_cast<int32_t*>(dp(cum_tiles)),
            static_cast<int32_t*>(dp(tile_offsets)), static_cast<int32_t*>(dp(tile_order)),
            static_cast<int64_t*>(dp(stats_dev)), (int)cap_box, static_cast<int32_t*>(dp(keep_scan)), dp(scratch),
            cap_listed, static_cast<int32_t*>(dp(flatten_ids)), static_cast<uint64_t*>(dp(sort_keys)), (int)seg_stride,
            ip(enum_order), static_cast<uint64_t*>(dp(isect_ids)), len_hint,
            reinterpret_cast<int64_t*>(static_cast<uintptr_t>(stats_row)), seq, pack ? fp(*pack_colors) : nullptr,
            (pack && pack_colors->dim() == 3) ? 1 : 0, pack ? (int)pack_ch : 0, fpw(records), tp(tuning), sp(stream));
    else
        rc = api.project_and_bin_speculative(
            (int)C, (int)N, fp(means), fp(quats), fp(scales), fp(viewmats), fp(Ks), fp(opac), opac.dim() == 2 ? 1 : 0,
            (int)width, (int)height, (float)eps2d, (float)near_plane, (float)far_plane, (float)radius_clip, (int)cull,
            static_cast<int32_t*>(dp(radii)), fpw(means2d), fpw(depths), fpw(conics),
            static_cast<int32_t*>(dp(tiles_per_gauss)), static_cast<int32_t*>(dp(cum_tiles)),
            static_cast<int32_t*>(dp(tile_offsets)), static_cast<int32_t*>(dp(tile_order)),
            static_cast<int64_t*>(dp(stats_dev)), (int)cap_box, static_cast<int32_t*>(dp(keep_scan)), dp(scratch),
            cap_listed, static_cast<int32_t*>(dp(flatten_ids)), static_cast<uint64_t*>(dp(sort_keys)),
            static_cast<uint64_t*>(dp(isect_ids)), len_hint,
            reinterpret_cast<int64_t*>(static_cast<uintptr_t>(stats_row)), seq, pack ? fp(*pack_colors) : nullptr,
            (pack && pack_colors->dim() == 3) ? 1 : 0, pack ? (int)pack_ch : 0, fpw(records), tp(tuning), sp(stream));
    if (rc != 0 && rc != 1) check(rc, "mobgs_project_and_bin_speculative");
    return {rc, {radii, means2d, depths, conics, tiles_per_gauss, cum_tiles, tile_offsets, keep_scan, flatten_ids},
            tile_order, isect_ids, records};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "mobgs_amd host fast path (see csrc/fastpath.cpp)";
    m.def("bind", &bind);
    m.def("prep_fwd", &prep_fwd);
    m.def("prep_bwd", &prep_bwd);
    m.def("raster_fwd", &raster_fwd);
    m.def("raster_bwd", &raster_bwd);
    m.def("raster_bwd_decode", &raster_bwd_decode);
    m.def("raster_bwd_decode_finish", &raster_bwd_decode_finish);
    m.def("raster_bwd_reduce_decode", &raster_bwd_reduce_decode);
    m.def("raster_bwd_reduce", &raster_bwd_reduce);
    m.def("decoder_fwd", &decoder_fwd);
    m.def("decoder_bwd", &decoder_bwd);
    m.def("project_bwd", &project_bwd);
    m.def("project_and_bin_speculative", &project_and_bin_speculative);
    m.def("project_prep_bwd", &project_prep_bwd);
    m.def("prep_state_buffers", &prep_state_buffers);
}
