#!/usr/bin/env python
"""Secondary metrics: train-mode render() and blurry-view throughput on one GPU.
    python scripts/bench_modes.py [--steps 20]
Prints one JSON object; not part of the bench.py contract."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import mobgs_amd.gaussian_renderer as GR  # noqa: E402
from mobgs_amd.distributed import SubframeShard  # noqa: E402


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def densify_legs(dev, steps, n=300_000):
    """per-iteration statistics and one densify_pruneclone on a 300k-splat table -- the fused path
    (mobgs_amd.densify) beside the reference's op sequence (torch index / cat / repeat per optimiser group,
    scene/gaussian_model.py:1044-1244, :1352-1356, :1480-1506) run with torch on the same GPU."""
    from mobgs_amd.densify import GROUPS, TrainableGaussians
    from mobgs_amd.synth import SynthCamera, dynamic_extras, gaussian_cloud

    class Opt:
        percent_dense, position_lr_init, feature_lr, featuret_lr, opacity_lr = 0.01, 0.00016, 0.0025, 0.001, 0.05
        scaling_lr, rotation_lr, omega_lr, zeta_lr, trbfc_lr, trbfs_lr, movelr, rgb_lr = \
            0.005, 0.001, 0.0001, 0.0001, 0.0001, 0.03, 3.5, 0.0001

    cam = SynthCamera(1352, 1014)
    p = gaussian_cloud(n, cam, 5)
    p.update(dynamic_extras(p["xyz"], 5))
    g = torch.Generator().manual_seed(1)
    p["scaling"] = p["scaling"] + 1.2 * torch.randn(n, 3, generator=g)
    vsp = (torch.randn(n, 2, generator=g) * 3e-4).to(dev)
    vis = (torch.rand(n, generator=g) > 0.4).to(dev)
    radii = torch.randint(0, 40, (n,), generator=g).to(torch.int32).to(dev)

    def fresh():
        pc = TrainableGaussians({k: p[k] for k in ("xyz", "scaling", "rotation", "opacity", "features_dc",
                                                    "features_t")},
                                {k: p[k] for k in ("omega", "trbf_center", "control_xyz", "current_control_num")},
                                device=dev)
        pc.training_setup(Opt())
        for gr in pc.optimizer.param_groups:
            for q in gr["params"]:
                if q.requires_grad and q.numel():
                    q.grad = torch.full_like(q, 1e-3)
        pc.optimizer.step()
        pc.add_densification_stats(vsp, vis, radii=radii)
        return pc

    out = {}
    pc = fresh()
    out["densify_stats_fused_ms"] = timed(lambda: pc.add_densification_stats(vsp, vis, radii=radii), steps)
    accum, denom, maxr = pc.xyz_gradient_accum, pc.denom, pc.max_radii2D
    rf = radii.float()

    def stats_torch():
        maxr[vis] = torch.max(maxr[vis], rf[vis])
        accum[vis] += torch.norm(vsp[vis, :2], dim=-1, keepdim=True)
        denom[vis] += 1

    out["densify_stats_torch_ops_ms"] = timed(stats_torch, steps)

    def fused_once():
        m = fresh()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.densify_pruneclone(2e-4, 0.005, 4.0, None, 2)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, m.get_xyz.shape[0]

    def torch_once():
        """the reference's sequence on plain tensors: clone (per-group index + cat, Adam moments cat zeros),
        split (per-group index + repeat + cat), prune (per-group boolean index incl. moments)"""
        m = fresh()
        st = m.table_state()
        names = [g for g, _ in GROUPS]
        P = {g: st[g].clone() for g in names}
        M = {g: (st[g + ".exp_avg"].clone(), st[g + ".exp_avg_sq"].clone()) for g in names if g + ".exp_avg" in st}
        table = st["_deformation_table"].clone()
        grads = (st["xyz_gradient_accum"] / st["denom"])
        grads[grads.isnan()] = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()

        def append(new, new_table):
            nonlocal table
            for g2 in names:
                P[g2] = torch.cat((P[g2], new[g2]), 0)
                if g2 in M:
                    M[g2] = (torch.cat((M[g2][0], torch.zeros_like(new[g2])), 0),
                             torch.cat((M[g2][1], torch.zeros_like(new[g2])), 0))
            table = torch.cat([table, new_table], -1)

        big = torch.max(torch.exp(P["scaling"]), dim=1).values > 0.01 * 4.0
        sel = torch.logical_and(torch.norm(grads, dim=-1) >= 2e-4, ~big)
        append({g2: P[g2][sel] for g2 in names}, table[sel])
        nn_ = P["xyz"].shape[0]
        padded = torch.zeros(nn_, device=dev)
        padded[:grads.shape[0]] = grads.squeeze()
        scale = torch.exp(P["scaling"])
        sel2 = torch.logical_and(padded >= 2e-4, torch.max(scale, dim=1).values > 0.01 * 4.0)
        stds = scale[sel2].repeat(2, 1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds)
        q = torch.nn.functional.normalize(P["rotation"][sel2])
        w, x, y, z = q.unbind(-1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                         1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                         1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3).repeat(2, 1, 1)
        new = {g2: P[g2][sel2].repeat(2, *([1] * (P[g2].dim() - 1))) for g2 in names}
        new["xyz"] = torch.bmm(R, samples.unsqueeze(-1)).squeeze(-1) + P["xyz"][sel2].repeat(2, 1)
        new["scaling"] = torch.log(scale[sel2].repeat(2, 1) / 1.6)
        append(new, table[sel2].repeat(2))
        keep = ~torch.cat((sel2, torch.zeros(2 * int(sel2.sum()), dtype=torch.bool, device=dev)))
        for g2 in names:
            P[g2] = P[g2][keep]
            if g2 in M:
                M[g2] = (M[g2][0][keep], M[g2][1][keep])
        table = table[keep]
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, P["xyz"].shape[0]

    fused_once()
    torch_once()
    f = [fused_once() for _ in range(3)]
    t = [torch_once() for _ in range(3)]
    assert f[0][1] == t[0][1], (f[0][1], t[0][1])
    out["densify_pruneclone_fused_ms"] = min(v[0] for v in f)
    out["densify_pruneclone_torch_ops_ms"] = min(v[0] for v in t)
    out["densify_rows_after"] = f[0][1]
    return out


def normals_legs(dev, steps, W, H):
    """get_normals (main_utils.py:95-141, train.py:590): the reference's op sequence (numpy direction map + H2D
    copy + torch ops, per call) beside the fused kernel, forward only (train.py never back-propagates through it)."""
    import types

    import numpy as np

    from mobgs_amd.main_utils import get_normals
    meta = types.SimpleNamespace(scale_factor_x=1170.0, scale_factor_y=1170.0, principal_point_x=W / 2,
                                 principal_point_y=H / 2, skew=0.0, use_center=True)
    z = (2.0 + torch.rand(1, H, W)).to(dev)

    def reference_ops():
        xx, yy = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
        pixels = np.stack([xx, yy], axis=-1) + 0.5
        y = (pixels[..., 1] - meta.principal_point_y) / meta.scale_factor_y
        x = (pixels[..., 0] - meta.principal_point_x - y * meta.skew) / meta.scale_factor_x
        viewdirs = torch.from_numpy(np.stack([x, y, np.ones_like(x)], axis=-1)).to(dev)
        coords = (viewdirs[None] * z[..., None]).squeeze(0)
        hd, wd, _ = coords.shape
        n = torch.cross(coords[1:hd - 1, 2:wd] - coords[1:hd - 1, 0:wd - 2],
                        coords[0:hd - 2, 1:wd - 1] - coords[2:hd, 1:wd - 1], dim=-1)
        n = torch.nn.functional.normalize(n, p=2, dim=-1)
        return torch.nn.functional.pad(n.permute(2, 0, 1), (1, 1, 1, 1), mode="constant")[None]

    return {"get_normals_reference_ops_ms": timed(reference_ops, max(3, steps // 4)),
            "get_normals_fused_ms": timed(lambda: get_normals(z, meta), steps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    W, H = 1352, 1014
    scam, cam, stat, dyn, _ = B.build_scene(dev, 200_000, 100_000, W, H)
    bg = torch.zeros(9, device=dev)
    g = torch.Generator().manual_seed(100)
    v3 = torch.randn(3, H, W, generator=g).to(dev)
    v1 = torch.randn(1, H, W, generator=g).to(dev)
    params = B.leaves(stat, dyn)

    def zero():
        for p in params:
            p.grad = None

    def lean():
        zero()
        out = GR.render(cam, stat, dyn, None, bg)
        torch.autograd.backward([out["render"], out["depth"]], [v3, v1])

    def train_mode():
        zero()
        out = GR.render(cam, stat, dyn, None, bg, get_static=True, get_dynamic=True)
        torch.autograd.backward([out["render"], out["depth"], out["s_render"], out["d_render"], out["d_alpha"],
                                 out["s_alpha"], out["d_depth"]], [v3, v1, v3, v3, v1, v1, v1])

    deltas = torch.linspace(-0.4, 0.4, 9).to(dev)
    shard = SubframeShard(1, 0)

    def blurry_view():
        """train.py:441-541 for ONE view: mid render in train mode + 8 latent renders (the reference also asks for
        static/dynamic outputs there, :512-516, and uses only render/depth) -> mean -> backward."""
        zero()
        mid = GR.render(cam, stat, dyn, None, bg, get_static=True, get_dynamic=True)

        def unit(k):
            if k == 4:
                return mid["render"]
            return GR.render(cam, stat, dyn, None, bg, get_static=True, get_dynamic=True,
                             delta_exposure=deltas[k])["render"]

        pred = shard.render_blurry_view(unit, 9)
        torch.autograd.backward([pred, mid["depth"], mid["d_alpha"]], [v3, v1, v1])

    # the same with BLCE-warped latent cameras and exposure offsets (train.py:472, 502-541): pose gradients flow
    # through the warped cameras (w2c of the rasterizer AND the decoder's view rays) into the BLCE parameters
    from mobgs_amd.blce import blceKernel
    cam.uid = 0
    cam.image = torch.rand(3, H, W, generator=g).to(dev)
    kern = blceKernel(num_views=2, num_warp=9, iteration=10000).to(dev)

    def blurry_view_blce():
        zero()
        kern.optimizer.zero_grad(set_to_none=True)
        cams, expo = kern.get_warped_cams(cam, cam, cam)
        mid = GR.render(cam, stat, dyn, None, bg, get_static=True, get_dynamic=True)

        def unit(k):
            if k == 4:
                return mid["render"]
            return GR.render(cams[k], stat, dyn, None, bg, get_static=True, get_dynamic=True,
                             delta_exposure=expo[k])["render"]

        pred = shard.render_blurry_view(unit, 9)
        torch.autograd.backward([pred, mid["depth"], mid["d_alpha"]], [v3, v1, v1])

    # photometric loss of one 1352x1014 view (train.py:621-628), fwd + bwd: torch ops as the reference calls them
    # vs the fused kernels
    from mobgs_amd.loss_utils import photometric_loss
    import torch.nn.functional as F

    def ref_ops_loss(x, y):
        """the op sequence of utils/loss_utils.py:233-239, 351-381 (L1 + 11x11 Gaussian-window SSIM as six depthwise
        convolutions), run with torch on the GPU: timing baseline only"""
        k = torch.exp(-(torch.arange(11, dtype=torch.float32, device=x.device) - 5) ** 2 / (2 * 1.5 ** 2))
        k = k / k.sum()
        win = (k[:, None] * k[None, :])[None, None].expand(3, 1, 11, 11).contiguous()
        conv = lambda t: F.conv2d(t, win, padding=5, groups=3)  # noqa: E731
        mx, my = conv(x), conv(y)
        sx, sy, sxy = conv(x * x) - mx * mx, conv(y * y) - my * my, conv(x * y) - mx * my
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        ssim_map = ((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sx + sy + c2))
        return (x - y).abs().mean() + 0.2 * (1.0 - ssim_map.mean())

    gt_img = torch.rand(1, 3, H, W, generator=g).to(dev)
    pred_img = torch.rand(1, 3, H, W, generator=g).to(dev).requires_grad_(True)

    def loss_torch_ops():
        pred_img.grad = None
        ref_ops_loss(pred_img, gt_img).backward()

    def loss_fused():
        pred_img.grad = None
        photometric_loss(pred_img, gt_img, 0.2).backward()

    v2 = torch.randn(1, H, W, 2, generator=g).to(dev)

    def flow_call():
        """One of the 9 get_flow() calls per view (train.py:570-579), forward + backward."""
        zero()
        e2m, m2e, img, alpha = GR.get_flow(cam, stat, dyn, None, bg, delta_exposure=deltas[1])
        torch.autograd.backward([e2m, m2e, img, alpha], [v2, v2, v3, v1])

    z2, z3, z1 = torch.zeros_like(v2), torch.zeros_like(v3), torch.zeros_like(v1)

    def flow_call_zero_weight():
        """the same with lambda_flow_loss = 0 (arguments/stereo/seesaw.py): the flow loss is still formed and
        back-propagated, but every cotangent is exactly zero -- the compositor's backward skips such pixels"""
        zero()
        e2m, m2e, img, alpha = GR.get_flow(cam, stat, dyn, None, bg, delta_exposure=deltas[1])
        torch.autograd.backward([e2m, m2e, img, alpha], [z2, z2, z3, z1])

    def flow_many_zero_weight():
        """all 9 get_flow calls of one view through get_flow_many (shared mid-exposure state), weight 0"""
        zero()
        outs = GR.get_flow_many(cam, stat, dyn, None, bg, [deltas[k] for k in range(9)])
        torch.autograd.backward([t for o in outs for t in o], [z2, z2, z3, z1] * 9)

    res = {}
    res["get_flow_x9_many_zero_weight_ms"] = timed(flow_many_zero_weight, max(3, a.steps // 4), warmup=1)
    res["get_flow_ms"] = timed(flow_call, a.steps)
    res["get_flow_zero_weight_ms"] = timed(flow_call_zero_weight, a.steps)
    res["photo_loss_torch_ops_ms"] = timed(loss_torch_ops, a.steps)
    res["photo_loss_fused_ms"] = timed(loss_fused, a.steps)
    res["lean_ms"] = timed(lean, a.steps)
    GR.INKERNEL_RAYS = False
    res["blurry_view_blce_ray_maps_ms"] = timed(blurry_view_blce, max(3, a.steps // 4), warmup=1)
    GR.INKERNEL_RAYS = True
    res["blurry_view_blce_inkernel_rays_ms"] = timed(blurry_view_blce, max(3, a.steps // 4), warmup=1)
    GR.FUSE_LAYERS = False
    res["train_mode_separate_passes_ms"] = timed(train_mode, a.steps)
    res["blurry_view_separate_passes_ms"] = timed(blurry_view, max(3, a.steps // 4), warmup=1)
    GR.FUSE_LAYERS = True
    res["train_mode_layered_ms"] = timed(train_mode, a.steps)
    res["blurry_view_layered_ms"] = timed(blurry_view, max(3, a.steps // 4), warmup=1)

    from mobgs_amd.ops import LeafGradSink

    def blurry_view_sink():
        """the same with the 9 renders' leaf gradients accumulated in-kernel (LeafGradSink around backward)"""
        zero()
        mid = GR.render(cam, stat, dyn, None, bg, get_static=True, get_dynamic=True)

        def unit(k):
            if k == 4:
                return mid["render"]
            return GR.render(cam, stat, dyn, None, bg, get_static=True, get_dynamic=True,
                             delta_exposure=deltas[k])["render"]

        pred = shard.render_blurry_view(unit, 9)
        with LeafGradSink(stat, dyn):
            torch.autograd.backward([pred, mid["depth"], mid["d_alpha"]], [v3, v1, v1])

    res["blurry_view_layered_gradsink_ms"] = timed(blurry_view_sink, max(3, a.steps // 4), warmup=1)
    res["train_mode_renders_per_s"] = 1e3 / res["train_mode_layered_ms"]
    res["blurry_views_per_s"] = 1e3 / res["blurry_view_layered_ms"]
    res.update(densify_legs(dev, a.steps))
    res.update(normals_legs(dev, a.steps, W, H))
    res["config"] = "seesaw-synth 200k+100k, 1352x1014, K=9 (1 GPU)"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
