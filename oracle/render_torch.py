"""CPU ORACLE (test infrastructure, NOT product code) -- plain-PyTorch restatement of the reference's render
glue around the rasterizer:

    interpolate_cubic_hermite   /root/reference/gaussian_renderer/__init__.py:23-56
    render()                    /root/reference/gaussian_renderer/__init__.py:59-316
    get_flow()                  /root/reference/gaussian_renderer/__init__.py:318-492
    get_flow_static()           /root/reference/gaussian_renderer/__init__.py:494-552
    Sandwich.forward            /root/reference/helper_model.py:19-28

PINNED: tests/test_oracle_cpu.py checks this module against fixtures under tests/golden/ produced by running
the reference's own functions in this container (tests/golden/make_golden.py) -- and against the reference
directly when /root/reference is present.  The rasterizer underneath is oracle/gsplat_torch.py (PARITY UNPINNED
vs. real gsplat, see there).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import gsplat_torch as G


def hermite(control_xyz: torch.Tensor, t: torch.Tensor, n_ctrl: torch.Tensor) -> torch.Tensor:
    """control_xyz [Nd,12,3]; t scalar tensor in [0,1]; n_ctrl [Nd,1] int64 (active knots per splat) -> [Nd,3]."""
    n = n_ctrl.reshape(-1)  # [Nd]
    ts = t * (n - 1).to(control_xyz.dtype)
    i = torch.clamp(torch.floor(ts).long(), torch.zeros_like(n), n - 2)
    il = torch.clamp(i - 1, torch.zeros_like(n), n - 1)
    ir = torch.clamp(i + 1, torch.zeros_like(n), n - 1)
    irr = torch.clamp(i + 2, torch.zeros_like(n), n - 1)
    u = (ts - i.to(ts.dtype))[:, None]

    def pick(idx):
        return torch.gather(control_xyz, 1, idx[:, None, None].expand(-1, 1, 3)).squeeze(1)

    p0, p1, p2, p3 = pick(il), pick(i), pick(ir), pick(irr)
    m0 = torch.where((il == i)[:, None], p2 - p1, (p2 - p0) / 2)
    m1 = torch.where((irr == ir)[:, None], p2 - p1, (p3 - p1) / 2)
    h00 = (1 + 2 * u) * (1 - u) ** 2
    h10 = u * (1 - u) ** 2
    h01 = u ** 2 * (3 - 2 * u)
    h11 = u ** 2 * (u - 1)
    return h00 * p1 + h10 * m0 + h01 * p2 + h11 * m1


def sandwich(w1: torch.Tensor, w2: torch.Tensor, feat: torch.Tensor, rays: torch.Tensor) -> torch.Tensor:
    """feat [1,9,H,W], rays [1,6,H,W], w1 [6,12(,1,1)], w2 [3,6(,1,1)] -> [1,3,H,W]."""
    albedo, spec, tf = feat.chunk(3, dim=1)
    x = torch.cat([spec, tf, rays], dim=1)
    y = torch.einsum("oc,bchw->bohw", w1.reshape(6, 12), x)
    y = torch.einsum("oc,bchw->bohw", w2.reshape(3, 6), F.relu(y))
    return torch.sigmoid(albedo + y)


def _decoder_weights(pc):
    dec = pc.rgbdecoder
    return dec.mlp1.weight, dec.mlp2.weight


def _time_tensor(x, like):
    return torch.as_tensor(x, dtype=like.dtype)


def _dyn_state(dyn_pc, time, max_time, delta):
    """tforpoly [Nd,1], quats (normalised) [Nd,4], means [Nd,3], colors [Nd,9] at time (+delta/max_time)."""
    trbf = dyn_pc.get_trbfcenter
    ones = torch.ones((dyn_pc.get_xyz.shape[0], 1), dtype=dyn_pc.get_xyz.dtype)
    if delta is not None:
        tt = time + delta / max_time
        curr = torch.clamp(_time_tensor(time, ones) + delta / max_time, 0, 1)
    else:
        tt = time
        curr = _time_tensor(time, ones)
    tfp = (tt * ones - trbf).detach()
    quats = F.normalize(dyn_pc._rotation + tfp * dyn_pc._omega)
    means = hermite(dyn_pc.get_control_xyz, curr, dyn_pc.current_control_num) * 1e-2
    colors = torch.cat((dyn_pc._features_dc, tfp * dyn_pc._features_t), dim=1)
    return tfp, quats, means, colors


def render(cam, stat_pc, dyn_pc, bg_color, get_static=False, get_dynamic=False, w2c=None, delta_exposure=None,
           get_flow=False, rasterization=G.rasterization, fully_fused_projection=G.fully_fused_projection):
    W, H = int(cam.image_width), int(cam.image_height)
    viewmat = cam.world_view_transform.transpose(0, 1) if w2c is None else w2c
    K = cam.K
    bg = torch.cat([bg_color[:3]] * 3, dim=-1)
    w1, w2 = _decoder_weights(dyn_pc)

    s_means, s_scales = stat_pc.get_xyz, stat_pc.get_scaling
    s_quats, s_opac, s_cols = stat_pc.get_rotation_stat, stat_pc.get_opacity, stat_pc.get_features_static
    _, d_quats, d_means, d_cols = _dyn_state(dyn_pc, cam.time, cam.max_time, delta_exposure)
    d_scales, d_opac = torch.exp(dyn_pc._scaling), dyn_pc.get_opacity

    def raster(means, quats, scales, opac, cols, bgs, mode):
        return rasterization(means=means, quats=quats, scales=scales, opacities=opac.squeeze(-1), colors=cols,
                             backgrounds=bgs, viewmats=viewmat[None], Ks=K[None], width=W, height=H, packed=False,
                             render_mode=mode)

    def decode(img):
        return sandwich(w1, w2, img[..., :-1].permute(0, 3, 1, 2), cam.cam_ray).squeeze(0)

    out = {k: None for k in ("s_render", "s_depth", "d_render", "d_depth", "d_alpha", "d_means3d", "s_alpha",
                             "blending_factor", "world_coordinates", "splat_center", "ori_flow", "ori_coord_map",
                             "labels", "centroids")}
    if get_dynamic:
        d_img, _, _ = raster(d_means, d_quats, d_scales, d_opac, d_cols, bg[None], "RGB+ED")
        out["d_depth"] = d_img[..., -1]
        out["d_render"] = decode(d_img)
        d_a, _, _ = raster(d_means, d_quats, d_scales, d_opac, torch.ones(d_cols.shape[0], 1), bg[0:1][None], "RGB")
        out["d_alpha"] = d_a[..., 0]
        out["d_means3d"] = d_means

    means = torch.cat((s_means, d_means), 0)
    scales = torch.cat((s_scales, d_scales), 0)
    quats = torch.cat((s_quats, d_quats), 0)
    opac = torch.cat((s_opac, d_opac), 0)
    cols = torch.cat((s_cols, d_cols), 0)

    if delta_exposure is not None and get_flow:
        _, o_quats, o_means, _ = _dyn_state(dyn_pc, cam.time, cam.max_time, None)
        # NB the reference feeds the UN-normalised original-time rotation here (:98, :189); the kernel normalises
        o_quats_raw = dyn_pc._rotation + (cam.time * torch.ones_like(dyn_pc.get_trbfcenter)
                                          - dyn_pc.get_trbfcenter).detach() * dyn_pc._omega
        _, ori_m2d, _, _, _ = fully_fused_projection(means=torch.cat((s_means, o_means), 0), covars=None,
                                                     quats=torch.cat((s_quats, o_quats_raw), 0), scales=scales,
                                                     viewmats=viewmat[None], Ks=K[None], width=W, height=H)

    img, _, info = raster(means, quats, scales, opac, cols, bg[None], "RGB+ED")
    out["depth"] = img[..., -1]
    radii = info["radii"].squeeze(0)
    try:
        info["means2d"].retain_grad()
    except Exception:  # noqa: BLE001
        pass
    rendered = decode(img)
    out["render"] = rendered

    if get_static:
        s_img, _, _ = raster(s_means, s_quats, s_scales, s_opac, s_cols, bg[None], "RGB+ED")
        out["s_depth"] = rendered[..., -1]  # the reference slices the DECODED image here (:250) -> [3,H]
        out["s_render"] = decode(s_img)
        s_a, _, _ = raster(s_means, s_quats, s_scales, s_opac, torch.ones(s_cols.shape[0], 1), bg[0:1][None], "RGB")
        out["s_alpha"] = s_a[..., 0]

    if delta_exposure is not None and get_flow:
        flow_2d = (ori_m2d - info["means2d"].clone().detach()).squeeze(0)
        flow_img, _, _ = raster(means, quats, scales, opac, flow_2d, None, "RGB")
        out["ori_flow"] = flow_img
        out["ori_coord_map"] = torch.tensor(cam.get_pixels(W, H, use_center=False)).type_as(flow_img) + flow_img

    out.update({"viewspace_points": info["means2d"], "visibility_filter": radii > 0, "radii": radii,
                "means_3d_final": means * 1e2, "colors_precomp_final": cols, "means_3d": d_means})
    return out


def get_flow(cam, stat_pc, dyn_pc, bg_color, delta_exposure, rasterization=G.rasterization,
             fully_fused_projection=G.fully_fused_projection):
    W, H = int(cam.image_width), int(cam.image_height)
    viewmat = cam.world_view_transform.transpose(0, 1)
    K = cam.K
    bg = torch.cat([bg_color[:3]] * 3, dim=-1)
    w1, w2 = _decoder_weights(dyn_pc)
    s_means, s_scales = stat_pc.get_xyz, stat_pc.get_scaling
    s_quats, s_opac, s_cols = stat_pc.get_rotation_stat, stat_pc.get_opacity, stat_pc.get_features_static
    _, mid_q, mid_m, _ = _dyn_state(dyn_pc, cam.time, cam.max_time, None)
    # get_flow clamps the mid time too (:354); cam.time is already in [0,1]
    _, exp_q, exp_m, exp_c = _dyn_state(dyn_pc, cam.time, cam.max_time, delta_exposure)
    d_scales, d_opac = torch.exp(dyn_pc._scaling), dyn_pc.get_opacity

    def raster(means, quats, scales, opac, cols, bgs, mode):
        return rasterization(means=means, quats=quats, scales=scales, opacities=opac.squeeze(-1), colors=cols,
                             backgrounds=bgs, viewmats=viewmat[None], Ks=K[None], width=W, height=H, packed=False,
                             render_mode=mode)

    latent_alpha, _, _ = raster(exp_m, exp_q, d_scales, d_opac, torch.ones(exp_c.shape[0], 1), bg[0:1][None], "RGB")
    latent_alpha = latent_alpha[..., 0]
    mid_means = torch.cat((s_means, mid_m), 0)
    mid_quats = torch.cat((s_quats, mid_q), 0)
    exp_means = torch.cat((s_means, exp_m), 0)
    exp_quats = torch.cat((s_quats, exp_q), 0)
    scales = torch.cat((s_scales, d_scales), 0)
    opac = torch.cat((s_opac, d_opac), 0)
    exp_cols = torch.cat((s_cols, exp_c), 0)
    _, mid_2d, _, _, _ = fully_fused_projection(means=mid_means, covars=None, quats=mid_quats, scales=scales,
                                                viewmats=viewmat[None], Ks=K[None], width=W, height=H)
    _, exp_2d, _, _, _ = fully_fused_projection(means=exp_means, covars=None, quats=exp_quats, scales=scales,
                                                viewmats=viewmat[None], Ks=K[None], width=W, height=H)
    e2m = (mid_2d - exp_2d).squeeze(0)
    pix = torch.tensor(cam.get_pixels(W, H, use_center=False))
    e2m_img, _, _ = raster(exp_means, exp_quats, scales, opac, e2m, None, "RGB")
    exp2mid = pix.type_as(e2m_img) + e2m_img
    m2e_img, _, _ = raster(mid_means, mid_quats, scales, opac, -e2m, None, "RGB")
    mid2exp = pix.type_as(m2e_img) + m2e_img
    latent, _, _ = raster(exp_means, exp_quats, scales, opac, exp_cols, bg[None], "RGB+ED")
    latent_img = sandwich(w1, w2, latent[..., :-1].permute(0, 3, 1, 2), cam.cam_ray).squeeze(0)
    return exp2mid, mid2exp, latent_img, latent_alpha


def get_flow_static(source_cam, target_cam, splat_cam, stat_pc, rasterization=G.rasterization,
                    fully_fused_projection=G.fully_fused_projection):
    s_means, s_scales = stat_pc.get_xyz, stat_pc.get_scaling
    s_quats, s_opac = stat_pc.get_rotation_stat, stat_pc.get_opacity
    K = source_cam.K

    def proj(cam):
        return fully_fused_projection(means=s_means, covars=None, quats=s_quats, scales=s_scales,
                                      viewmats=cam.world_view_transform.transpose(0, 1)[None], Ks=K[None],
                                      width=int(cam.image_width), height=int(cam.image_height))[1]

    flow_2d = (proj(source_cam) - proj(target_cam)).squeeze(0)
    img, _, _ = rasterization(means=s_means, quats=s_quats, scales=s_scales, opacities=s_opac.squeeze(-1),
                              colors=flow_2d, backgrounds=None,
                              viewmats=splat_cam.world_view_transform.transpose(0, 1)[None], Ks=K[None],
                              width=int(splat_cam.image_width), height=int(splat_cam.image_height), packed=False,
                              render_mode="RGB")
    return flow_2d, img


def psnr(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """/root/reference/utils/image_utils.py:17-31 (mask=None branch): per-image 20*log10(1/sqrt(mse))."""
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return (20 * torch.log10(1.0 / torch.sqrt(mse.float()))).mean().double()


def masked_l1(network_output: torch.Tensor, gt: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """/root/reference/utils/loss_utils.py:233-237 (the mask branch of l1_loss)."""
    mask = mask.expand(-1, gt.shape[1], -1, -1)
    return torch.abs((network_output - gt) * mask).sum() / (mask.sum() + 1e-8)


def flow_warp_loss(ori_image_tensor: torch.Tensor, latent_img_final_tensor: torch.Tensor,
                   exp2mid_coord_final_tensor: torch.Tensor, mid2exp_coord_final_tensor: torch.Tensor,
                   latent_alpha_final_tensor: torch.Tensor, d_alpha_tensor: torch.Tensor) -> torch.Tensor:
    """/root/reference/train.py:651-671 without the lambda_flow_loss factor, statement by statement (the reference
    normalises the coordinate tensors IN PLACE -- they are fresh torch.cat results there; here on clones).
    ori [B,3,H,W]; latent [B,K,3,H,W]; coords [B,K,H,W,2] in pixels; latent_alpha [B,K,1,H,W]; d_alpha [B,1,H,W].
    Parity anchor: these are the reference's own torch calls (F.grid_sample is the third-party piece, and it is torch
    itself), evaluated on the CPU."""
    K = latent_img_final_tensor.shape[1]
    H, W = ori_image_tensor.shape[-2:]

    def norm(coord):
        x = coord[..., 0] / (W - 1)
        y = coord[..., 1] / (H - 1)
        return (2.0 * torch.stack([x, y], dim=-1) - 1.0).flatten(0, 1)

    ori_k = ori_image_tensor.unsqueeze(1).expand(-1, K, -1, -1, -1).flatten(0, 1)
    warped_exp2mid = F.grid_sample(ori_k, norm(exp2mid_coord_final_tensor), mode='bilinear', padding_mode='border',
                                   align_corners=False).reshape(-1, K, 3, H, W)
    warped_mid2exp = F.grid_sample(latent_img_final_tensor.flatten(0, 1), norm(mid2exp_coord_final_tensor),
                                   mode='bilinear', padding_mode='border', align_corners=False).reshape(-1, K, 3, H, W)
    return masked_l1(warped_exp2mid.flatten(0, 1), latent_img_final_tensor.flatten(0, 1),
                     latent_alpha_final_tensor.flatten(0, 1)) \
        + masked_l1(warped_mid2exp.flatten(0, 1), ori_k,
                    d_alpha_tensor.unsqueeze(1).expand(-1, K, -1, -1, -1).flatten(0, 1))
