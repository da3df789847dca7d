#!/usr/bin/env python
"""A miniature of the reference's training iteration (train.py:441-820) on a synthetic scene, end to end on the
MI355X path: render() of a static + a dynamic Gaussian set -> photometric loss (L1 + 0.2 D-SSIM) -> backward ->
densification statistics -> Adam step; every `densify_every` iterations densify_pruneclone / prune / opacity reset.

    python examples/train_synth.py [--iters 200] [--ns 20000] [--nd 10000] [--width 676 --height 507]

The "ground truth" is a render of the same scene with perturbed parameters, so the loss has somewhere to go.
Returns (and prints) the loss / PSNR trajectory; used by tests/test_gpu_train_loop.py as an integration test.
"""
import argparse
import os
import sys

import torch

torch.autograd.set_multithreading_enabled(False)  # backward on the calling thread: no device-thread wake-up per step

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobgs_amd.camera import PinholeCamera  # noqa: E402
from mobgs_amd.densify import TrainableGaussians  # noqa: E402
from mobgs_amd.gaussian_renderer import render  # noqa: E402
from mobgs_amd.helper_model import Sandwich  # noqa: E402
from mobgs_amd.loss_utils import photometric_loss, psnr  # noqa: E402
from mobgs_amd.synth import SynthCamera, dynamic_extras, gaussian_cloud  # noqa: E402


class Opt:
    """OptimizationParams defaults of the reference (arguments/__init__.py:117-185) that the loop reads."""
    percent_dense = 0.01
    position_lr_init = 0.00016
    feature_lr = 0.0025
    featuret_lr = 0.001
    opacity_lr = 0.05
    scaling_lr = 0.005
    rotation_lr = 0.001
    omega_lr = 0.0001
    zeta_lr = 0.0001
    trbfc_lr = 0.0001
    trbfs_lr = 0.03
    movelr = 3.5
    rgb_lr = 0.0001
    opthr = 0.005
    densify_grad_threshold = 0.0002
    lambda_dssim = 0.2


def build(dev, ns, nd, width, height, seed=0):
    scam = SynthCamera().scaled(width, height)
    torch.manual_seed(seed)
    dec = Sandwich(9, 3).to(dev)
    sp, dp = gaussian_cloud(ns, scam, seed), gaussian_cloud(nd, scam, seed + 1)
    dx = dynamic_extras(dp["xyz"], seed)
    stat = TrainableGaussians(sp, None, dec, device=dev)
    dyn = TrainableGaussians(dp, dx, dec, device=dev)
    cams = [PinholeCamera(width, height, scam.K, torch.eye(4), time=t / 23.0, max_time=23, device=dev)
            for t in (5.0, 11.0, 17.0)]
    return stat, dyn, cams, (sp, dp, dx)


def train(dev="cuda:0", iters=120, ns=6000, nd=3000, width=320, height=240, densify_every=40, seed=0, log=None):
    stat, dyn, cams, (sp, dp, dx) = build(dev, ns, nd, width, height, seed)
    bg = torch.zeros(9, device=dev)
    # targets: the same scene with shifted colours / opacities, rendered once
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        tsp = dict(sp, features_dc=sp["features_dc"] + 0.5 * torch.randn(sp["features_dc"].shape, generator=g),
                   opacity=sp["opacity"] + 0.5)
        tdp = dict(dp, features_dc=dp["features_dc"] + 0.5 * torch.randn(dp["features_dc"].shape, generator=g))
        tstat = TrainableGaussians(tsp, None, stat.rgbdecoder, device=dev)
        tdyn = TrainableGaussians(tdp, dx, stat.rgbdecoder, device=dev)
        targets = [render(c, tstat, tdyn, None, bg)["render"].clamp(0, 1).detach() for c in cams]
    opt = Opt()
    stat.training_setup(opt)
    dyn.training_setup(opt)
    dyn.optimizer.param_groups = [gr for gr in dyn.optimizer.param_groups if gr["name"] != "decoder"]  # shared decoder
    history = []
    for it in range(1, iters + 1):
        k = it % len(cams)
        out = render(cams[k], stat, dyn, None, bg)
        image = out["render"]
        loss = photometric_loss(image, targets[k], opt.lambda_dssim)
        loss.backward()
        with torch.no_grad():
            Ns = stat.get_xyz.shape[0]
            vgrad = out["viewspace_points"].grad[0]          # [Ns+Nd, 2]
            vis, radii = out["visibility_filter"], out["radii"]
            stat.add_densification_stats(vgrad[:Ns], vis[:Ns], radii=radii[:Ns])
            dyn.add_densification_stats(vgrad[Ns:], vis[Ns:], radii=radii[Ns:])
            history.append((float(loss), float(psnr(image.detach().clamp(0, 1)[None], targets[k][None]).mean()),
                            Ns, dyn.get_xyz.shape[0]))
        stat.optimizer.step()
        dyn.optimizer.step()
        stat.optimizer.zero_grad(set_to_none=True)
        dyn.optimizer.zero_grad(set_to_none=True)
        if it % densify_every == 0 and it < iters:
            extent = 4.0
            stat.densify_pruneclone(opt.densify_grad_threshold * 1e-2, opt.opthr, extent, 20)
            dyn.densify_pruneclone(opt.densify_grad_threshold, opt.opthr, extent, 2)
            stat.prune_points((stat.get_opacity < opt.opthr).squeeze())
            dyn.prune_points((dyn.get_opacity < opt.opthr).squeeze())
        if log and (it % log == 0 or it == 1):
            print(f"it {it:4d}  loss {history[-1][0]:.5f}  psnr {history[-1][1]:.2f} dB  "
                  f"splats {history[-1][2]} + {history[-1][3]}")
    return history, stat, dyn


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--ns", type=int, default=20000)
    ap.add_argument("--nd", type=int, default=10000)
    ap.add_argument("--width", type=int, default=676)
    ap.add_argument("--height", type=int, default=507)
    a = ap.parse_args()
    train(iters=a.iters, ns=a.ns, nd=a.nd, width=a.width, height=a.height, log=20)
