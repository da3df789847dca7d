"""The backward compositor with the gradient sums on the matrix pipe (csrc/raster_bwd_mfma.hip, MobgsTuning.bwd_mfma):
both arms -- 1 = one wave per tile (+ the four-wave team for heavy tiles), 2 = the team for every tile -- against the
torch-autograd oracle (oracle/gsplat_torch.py) and against the quadrant kernel (arm 0) on the same lists.

Tolerances: vs the oracle the ones of test_gpu_operator_parity.py (rtol 1e-3, 5e-4 x the tensor's scale); between the
arms only the summation order differs (fp32 MFMA is a chain of exact fmaf): 1e-4 of the tensor's maximum, observed
<= 2e-5 (scripts/check_bwd_mfma.py at 300 k splats)."""
import pytest
import torch

from mobgs_amd.synth import SynthCamera, splat_inputs

pytestmark = pytest.mark.gpu

NAMES = ["means", "quats", "scales", "opacities", "colors", "viewmats"]


def _grads(fn, s, dev, w, h, v_img, v_a, bg, **kw):
    t = {k: v.to(dev).clone().requires_grad_(k in NAMES) for k, v in s.items()}
    img, a, _ = fn(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], t["viewmats"], t["Ks"], w, h,
                   packed=False, backgrounds=bg.to(dev), **kw)
    torch.autograd.backward([img, a], [v_img.to(dev), v_a.to(dev)])
    return {k: t[k].grad.detach().cpu() for k in NAMES}


@pytest.mark.parametrize("n,w,h,channels,heavy", [(6000, 200, 136, 9, None), (6000, 200, 136, 9, 0), (2500, 100, 70, 3, None),
                                                   (40000, 1352 // 2, 1014 // 2, 9, None)])
def test_mfma_arms_match_oracle_and_quadrant_kernel(hip_device, n, w, h, channels, heavy):
    from mobgs_amd import rendering
    from mobgs_amd.rendering import rasterization
    from oracle import gsplat_torch as G
    from helpers import close
    cam = SynthCamera().scaled(w, h)
    s = splat_inputs(n, cam, 4, channels)
    g = torch.Generator().manual_seed(9)
    D = channels + 1
    v_img = torch.randn(1, h, w, D, generator=g)
    v_a = torch.randn(1, h, w, 1, generator=g)
    bg = torch.rand(1, channels, generator=g)
    old = (rendering.tuning.bwd_mfma, rendering.tuning.heavy_tile_len)
    res = {}
    try:
        if heavy is not None:
            rendering.tuning.heavy_tile_len = heavy  # 0: no heavy tiles even on this small grid -> the wave-per-tile path
        for arm in (0, 1, 2):
            rendering.tuning.bwd_mfma = arm
            res[arm] = _grads(rasterization, s, hip_device, w, h, v_img, v_a, bg, render_mode="RGB+ED")
    finally:
        rendering.tuning.bwd_mfma, rendering.tuning.heavy_tile_len = old
    for arm in (1, 2):
        for k in NAMES:
            m = float(res[0][k].abs().max())
            close(res[arm][k], res[0][k], 1e-4, 1e-4 * m + 1e-9, f"grad[{k}] arm {arm} vs quadrant kernel")
    if n <= 6000:  # the torch oracle evaluates every (pixel, splat) pair: small scenes only
        ref = _grads(G.rasterization, s, torch.device("cpu"), w, h, v_img, v_a, bg, render_mode="RGB+ED")
        for arm in (1, 2):
            for k in NAMES:
                scale = float(ref[k].abs().max())
                close(res[arm][k], ref[k], 1e-3, 5e-4 * scale + 1e-6, f"grad[{k}] arm {arm} vs torch oracle")


def test_mfma_arms_in_class_restricted_passes(hip_device):
    """Static-only / dynamic-only passes over the lists of the whole set (the train-mode render): the two classes own
    disjoint slots of ONE gradient-slot buffer, so a pass must leave the other class's slots alone."""
    from mobgs_amd import rendering
    from helpers import close
    n, w, h, Ns = 5000, 176, 120, 3100
    cam = SynthCamera().scaled(w, h)
    s = splat_inputs(n, cam, 7, 9)
    names = ["means", "quats", "scales", "opacities", "colors"]

    def grads():
        t = {k: v.to(hip_device).clone().requires_grad_(k in names) for k, v in s.items()}
        sp = rendering.SharedProjection(t["means"], t["quats"], t["scales"], t["opacities"], t["viewmats"], t["Ks"], w, h)
        renders, alphas = sp.composite_layers(t["colors"], Ns, want_all=False, want_static=True, want_dynamic=True)
        outs = [renders[1], renders[2], alphas[1], alphas[2], sp.class_alpha(Ns, 2)]
        g = torch.Generator().manual_seed(5)  # the same cotangents for every arm
        torch.autograd.backward(outs, [torch.randn(o.shape, generator=g).to(hip_device) for o in outs])
        return {k: t[k].grad.detach().cpu() for k in names}

    old = rendering.tuning.bwd_mfma
    res = {}
    try:
        for arm in (0, 1, 2):
            rendering.tuning.bwd_mfma = arm
            res[arm] = grads()
    finally:
        rendering.tuning.bwd_mfma = old
    for arm in (1, 2):
        for k in names:
            m = float(res[0][k].abs().max())
            close(res[arm][k], res[0][k], 1e-4, 1e-4 * m + 1e-9, f"grad[{k}] arm {arm} vs quadrant kernel (class passes)")
