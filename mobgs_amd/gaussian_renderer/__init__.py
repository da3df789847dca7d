"""Render API (boundary B1): drop-in for the reference's `gaussian_renderer` package.

    from mobgs_amd.gaussian_renderer import render, get_flow, get_flow_static

mirror /root/reference/gaussian_renderer/__init__.py:59-316 (render), :318-492 (get_flow), :494-552
(get_flow_static): same positional/keyword arguments, same 22-key result dict (including the `s_depth` quirk
of :250), same autograd contract (`viewspace_points` is a non-leaf with retain_grad(), gradients reach every
Gaussian leaf, the decoder weights, `w2c` / the camera's world_view_transform and cam_ray).

Everything per-Gaussian and per-pixel runs in libmobgs_hip.so:
    prep (Hermite spline, rotation, activations, colour features, static|dynamic concat)  -> csrc/prep.hip
    projection / tile lists / compositing                                                 -> mobgs_amd.rendering
    expected-depth normalisation + Sandwich decoder                                       -> csrc/decoder.hip
Unused reference kwargs (pipe, scaling_modifier, override_color, stage, cam_type, ...) are accepted and ignored,
exactly as the reference ignores them.
"""
from __future__ import annotations

import torch

from .. import rendering as _R
from ..ops import PrepSplats, decode
from . import network_gui  # noqa: F401  (train.py imports it from here)

__all__ = ["render", "get_flow", "get_flow_static", "interpolate_cubic_hermite", "network_gui"]

# True: train-mode render() composites its three splat sets in one layered pass (csrc/raster_layers.hip);
# False: one rasterization per set, call for call like the reference (kept for A/B tests)
FUSE_LAYERS = True


def _device_of(pc):
    return pc.get_xyz.device


def _times(cam, delta_exposure, dev):
    """[t_feat, t_curve] on the device, without a host sync when delta_exposure is a device tensor."""
    t = torch.as_tensor(float(cam.time), dtype=torch.float32, device=dev)
    if delta_exposure is not None:
        d = delta_exposure if torch.is_tensor(delta_exposure) else torch.as_tensor(float(delta_exposure))
        t = t + d.detach().to(device=dev, dtype=torch.float32).reshape(()) / cam.max_time
    return torch.stack([t, torch.clamp(t, 0.0, 1.0)])


def _prep(stat_pc, dyn_pc, times):
    return PrepSplats.apply(times, stat_pc._xyz, stat_pc._scaling, stat_pc._rotation, stat_pc._opacity,
                            stat_pc._features_dc, stat_pc._features_t, dyn_pc.get_control_xyz,
                            dyn_pc.current_control_num, dyn_pc._scaling, dyn_pc._rotation, dyn_pc._omega,
                            dyn_pc._opacity, dyn_pc._features_dc, dyn_pc._features_t, dyn_pc.get_trbfcenter)


def _decoder_weights(dyn_pc):
    dec = dyn_pc.rgbdecoder
    return dec.mlp1.weight, dec.mlp2.weight


def interpolate_cubic_hermite(signal, times, N):
    """API of /root/reference/gaussian_renderer/__init__.py:23-56: signal [Nd,3,K], times [Nd,3,1], N [Nd,1].
    (Convenience wrapper; render() evaluates the spline inside the fused prep kernel.)"""
    ctrl = signal.permute(0, 2, 1).contiguous()
    nd = ctrl.shape[0]
    if ctrl.shape[1] != 12:
        raise NotImplementedError("the fused spline kernel is built for 12 control points (control_num = 12)")
    dev = ctrl.device
    t = times.reshape(nd, -1)[0, 0].to(torch.float32)
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)  # noqa: E731
    means, _, _, _, _ = PrepSplats.apply(torch.stack([t, t]), z(0, 3), z(0, 3), z(0, 4), z(0, 1), z(0, 6), z(0, 3),
                                         ctrl, N, z(nd, 3), z(nd, 4), z(nd, 4), z(nd, 1), z(nd, 6), z(nd, 3),
                                         z(nd, 1))
    return means * 1e2


def render(viewpoint_camera, stat_pc, dyn_pc, pipe, bg_color, scaling_modifier=1.0, override_color=None,
           stage="fine", cam_type=None, is_static=False, over_t=None, over_vde=None, get_static=False,
           get_dynamic=False, stat_stat=True, ref_wc=None, iter_fact=1, flow=None, coherent=None, target_ts=None,
           target_w2cs=None, get_heatmap=False, w2c=None, delta_exposure=None, get_flow=False, cluster=None):
    """Render the scene (see module docstring).  Returns the reference's result dict."""
    if cluster is not None:
        raise NameError("name 'labels' is not defined")  # the reference raises exactly this (:314-315)
    cam = viewpoint_camera
    dev = _device_of(dyn_pc)
    W, H = int(cam.image_width), int(cam.image_height)
    viewmat = cam.world_view_transform.transpose(0, 1) if w2c is None else w2c
    K = cam.K
    bg = torch.cat([bg_color[:3]] * 3, dim=-1)
    w1, w2 = _decoder_weights(dyn_pc)
    Ns = stat_pc.get_xyz.shape[0]

    dyn_sl, stat_sl, all_sl = slice(Ns, None), slice(0, Ns), slice(None)
    times = _times(cam, delta_exposure, dev)
    means, quats, scales, opac, cols = _prep(stat_pc, dyn_pc, times)
    if coherent is not None:
        means = torch.cat((means[:Ns], means[Ns:] + coherent), 0)

    def pick(t, sl):
        return t if sl is all_sl else t[sl]  # a full slice would still cost a zeros+copy pair in backward

    def raster(sl, colors, bgs, mode):
        return _R.rasterization(means=pick(means, sl), quats=pick(quats, sl), scales=pick(scales, sl),
                                opacities=pick(opac, sl), colors=colors, backgrounds=bgs, viewmats=viewmat[None],
                                Ks=K[None], width=W, height=H, packed=False, render_mode=mode)

    def decode_ed(img, alphas):
        return decode(img, alphas, cam.cam_ray, w1, w2, True)  # views only: no select/zeros/copy in backward

    out = {k: None for k in ("s_render", "s_depth", "d_render", "d_depth", "d_alpha", "d_means3d", "s_alpha",
                             "blending_factor", "world_coordinates", "splat_center", "ori_flow", "ori_coord_map",
                             "labels", "centroids")}

    ori_m2d = None
    if delta_exposure is not None and get_flow:
        o_means, o_quats, _, _, _ = _prep(stat_pc, dyn_pc, _times(cam, None, dev))
        _, ori_m2d, _, _, _ = _R.fully_fused_projection(means=o_means, covars=None, quats=o_quats, scales=scales,
                                                        viewmats=viewmat[None], Ks=K[None], width=W, height=H)

    layered = (get_static or get_dynamic) and FUSE_LAYERS
    if layered:
        # ONE projection, ONE tile binning / sort and ONE compositing walk for the combined, static-only and
        # dynamic-only renders (the reference: 5 rasterizations, :143-176, :201-214, :236-268)
        imgs, alps, info = _R.rasterize_layers(means, quats, scales, opac, cols, viewmat[None], K[None], W, H, Ns,
                                               backgrounds=bg[None], want_static=get_static,
                                               want_dynamic=get_dynamic)
        img, alphas = imgs[0], alps[0]
    else:
        img, alphas, info = _raster_acc(raster, all_sl, cols, bg[None])

    def alpha_pass(a):
        """The reference's ones-colour pass (:163-177, :255-269): sum_i w_i + T_final * bg = (1 - T) + T * bg."""
        return a + (1.0 - a) * bg[0]

    if get_dynamic:
        if layered:
            d_img, d_a = imgs[2], alps[2]
            out["d_alpha"] = alpha_pass(d_a)
        else:
            d_img, d_a, _ = _raster_acc(raster, dyn_sl, cols[dyn_sl], bg[None])
            ones = torch.ones(cols.shape[0] - Ns, 1, device=dev)
            out["d_alpha"] = raster(dyn_sl, ones, bg[0:1][None], "RGB")[0][..., 0]
        out["d_render"], d_depth = decode_ed(d_img, d_a)
        out["d_depth"] = d_depth.unsqueeze(0)
        out["d_means3d"] = means[dyn_sl]

    radii = info["radii"].squeeze(0)
    try:
        info["means2d"].retain_grad()
    except Exception:  # noqa: BLE001  (no grad mode)
        pass
    rendered, depth = decode_ed(img, alphas)
    out["render"] = rendered
    out["depth"] = depth.unsqueeze(0)

    if get_static:
        if layered:
            s_img, s_a = imgs[1], alps[1]
            out["s_alpha"] = alpha_pass(s_a)
        else:
            s_img, s_a, _ = _raster_acc(raster, stat_sl, cols[stat_sl], bg[None])
            ones = torch.ones(Ns, 1, device=dev)
            out["s_alpha"] = raster(stat_sl, ones, bg[0:1][None], "RGB")[0][..., 0]
        out["s_render"], _ = decode_ed(s_img, s_a)
        out["s_depth"] = rendered[..., -1]  # reference quirk (:250): slices the decoded image -> [3,H]

    if ori_m2d is not None:
        flow_2d = (ori_m2d - info["means2d"].detach()).squeeze(0)
        flow_img = raster(all_sl, flow_2d, None, "RGB")[0]
        out["ori_flow"] = flow_img
        out["ori_coord_map"] = _pixel_grid(cam, W, H, flow_img) + flow_img

    out.update({"viewspace_points": info["means2d"], "visibility_filter": radii > 0, "radii": radii,
                "means_3d_final": means * 1e2, "colors_precomp_final": cols, "means_3d": means[dyn_sl]})
    return out


def _raster_acc(raster, sl, colors, bgs):
    """'RGB+D' compositing (9 feature channels + accumulated depth); the ED division happens in the decoder."""
    img, alphas, info = raster(sl, colors, bgs, "RGB+D")
    return img, alphas.squeeze(-1), info


def _pixel_grid(cam, W, H, like):
    return torch.tensor(cam.get_pixels(W, H, use_center=False)).type_as(like)


def get_flow(viewpoint_camera, stat_pc, dyn_pc, pipe, bg_color, delta_exposure=None):
    """/root/reference/gaussian_renderer/__init__.py:318-492 ->
    (exp2mid_coord_map [1,H,W,2], mid2exp_coord_map [1,H,W,2], latent_img [3,H,W], latent_alpha [1,H,W])."""
    cam = viewpoint_camera
    dev = _device_of(dyn_pc)
    W, H = int(cam.image_width), int(cam.image_height)
    viewmat = cam.world_view_transform.transpose(0, 1)
    K = cam.K
    bg = torch.cat([bg_color[:3]] * 3, dim=-1)
    w1, w2 = _decoder_weights(dyn_pc)
    Ns = stat_pc.get_xyz.shape[0]
    mid_m, mid_q, scales, opac, _ = _prep(stat_pc, dyn_pc, _times(cam, None, dev))
    exp_m, exp_q, _, _, exp_c = _prep(stat_pc, dyn_pc, _times(cam, delta_exposure, dev))

    def raster(m, q, sl, colors, bgs, mode):
        return _R.rasterization(means=m[sl], quats=q[sl], scales=scales[sl], opacities=opac[sl], colors=colors,
                                backgrounds=bgs, viewmats=viewmat[None], Ks=K[None], width=W, height=H, packed=False,
                                render_mode=mode)

    def project(m, q):
        return _R.fully_fused_projection(means=m, covars=None, quats=q, scales=scales, viewmats=viewmat[None],
                                         Ks=K[None], width=W, height=H)[1]

    dyn_sl, all_sl = slice(Ns, None), slice(None)
    ones = torch.ones(exp_c.shape[0] - Ns, 1, device=dev)
    latent_alpha = raster(exp_m, exp_q, dyn_sl, ones, bg[0:1][None], "RGB")[0][..., 0]
    e2m = (project(mid_m, mid_q) - project(exp_m, exp_q)).squeeze(0)
    e2m_img = raster(exp_m, exp_q, all_sl, e2m, None, "RGB")[0]
    pix = _pixel_grid(cam, W, H, e2m_img)
    exp2mid = pix + e2m_img
    mid2exp = pix + raster(mid_m, mid_q, all_sl, -e2m, None, "RGB")[0]
    img, alphas, _ = raster(exp_m, exp_q, all_sl, exp_c, bg[None], "RGB+D")
    latent_img, _ = decode(img, alphas, cam.cam_ray, w1, w2, True)
    return exp2mid, mid2exp, latent_img, latent_alpha


def get_flow_static(source_camera, target_camera, splat_camera, stat_pc, dyn_pc, pipe, bg_color):
    """/root/reference/gaussian_renderer/__init__.py:494-552 -> (flow_2d [Ns,2], rendered_flow [1,H,W,2])."""
    means, scales = stat_pc.get_xyz, stat_pc.get_scaling
    quats, opac = stat_pc._rotation, stat_pc.get_opacity.squeeze(-1)
    K = source_camera.K

    def project(cam):
        return _R.fully_fused_projection(means=means, covars=None, quats=quats, scales=scales,
                                         viewmats=cam.world_view_transform.transpose(0, 1)[None], Ks=K[None],
                                         width=int(cam.image_width), height=int(cam.image_height))[1]

    flow_2d = (project(source_camera) - project(target_camera)).squeeze(0)
    img = _R.rasterization(means=means, quats=quats, scales=scales, opacities=opac, colors=flow_2d,
                           backgrounds=None, viewmats=splat_camera.world_view_transform.transpose(0, 1)[None],
                           Ks=K[None], width=int(splat_camera.image_width), height=int(splat_camera.image_height),
                           packed=False, render_mode="RGB")[0]
    return flow_2d, img
