"""csrc/decoder.hip at ragged sizes against a plain PyTorch fp32 restatement of the same op
(/root/reference/helper_model.py:19-28 Sandwich.forward + the 'ED' depth normalisation of gsplat's rendering.py):
pixel counts around the 64-lane wave, the 256-thread workgroup and the grid-stride loop of the software-pipelined
backward pass (its prefetch reads from clamped addresses), both ray sources, with and without the depth channel,
with and without a depth cotangent, 9 / 10 / 12 channels-last channels."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_decoder(feat, alphas, rays6, w1, w2, has_depth):
    """feat [H,W,CF], rays6 [6,H,W] -> rgb [3,H,W], depth [H,W] | None"""
    x = torch.cat([feat[..., 3:9].permute(2, 0, 1), rays6], dim=0)                 # spec | timefeat | rays
    h = torch.relu(torch.einsum("jc,chw->jhw", w1, x))
    y = torch.einsum("oj,jhw->ohw", w2, h)
    rgb = torch.sigmoid(feat[..., 0:3].permute(2, 0, 1) + y)
    depth = feat[..., 9] / alphas.clamp(min=1e-10) if has_depth else None
    return rgb, depth


@pytest.mark.parametrize("H,W", [(1, 1), (1, 63), (1, 64), (5, 13), (3, 257), (9, 768 // 3 + 1), (37, 83), (64, 193)])
@pytest.mark.parametrize("ray_map", [True, False])
@pytest.mark.parametrize("has_depth,CF,depth_cot", [(True, 10, True), (True, 10, False), (False, 9, False),
                                                    (True, 12, True)])
def test_decoder_matches_torch_at_ragged_sizes(hip_device, H, W, ray_map, has_depth, CF, depth_cot):
    from mobgs_amd.camera import PinholeCamera
    from mobgs_amd.ops import decode
    dev = hip_device
    g = torch.Generator().manual_seed(H * 1000 + W + CF)
    feat = torch.randn(H, W, CF, generator=g).to(dev)
    alphas = (0.05 + 0.95 * torch.rand(H, W, generator=g)).to(dev)
    alphas[0, 0] = 0.0                                                               # the clamp(1e-10) branch
    w1 = (0.5 * torch.randn(6, 12, generator=g)).to(dev)
    w2 = (0.5 * torch.randn(3, 6, generator=g)).to(dev)
    K = torch.tensor([[50.0, 0, W / 2], [0, 48.0, H / 2], [0, 0, 1]])
    c2w = torch.eye(4)[:3, :].clone()
    c2w[:, 3] = torch.tensor([0.2, -0.3, 0.1])
    intr = torch.stack([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]).to(dev)
    rays = PinholeCamera.build_cam_ray_c2w(W, H, K.to(dev), c2w.to(dev)).detach()    # [1,6,H,W]
    cot = torch.randn(3, H, W, generator=g).to(dev)
    cot_d = torch.randn(H, W, generator=g).to(dev)

    res = {}
    for name in ("hip", "torch"):
        f = feat.clone().requires_grad_(True)
        a = alphas.clone().requires_grad_(True)
        u1, u2 = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
        r = rays.clone().requires_grad_(True)
        if name == "hip":
            src = r if ray_map else (intr, c2w.to(dev))
            rgb, depth = decode(f, a if has_depth else None, src, u1, u2, has_depth)
        else:
            rgb, depth = _torch_decoder(f, a, r[0], u1, u2, has_depth)
        loss = (rgb * cot).sum()
        if has_depth and depth_cot:
            loss = loss + (depth * cot_d).sum()
        loss.backward()
        res[name] = dict(rgb=rgb.detach(), depth=None if depth is None else depth.detach(), f=f.grad, w1=u1.grad,
                         w2=u2.grad, a=a.grad, r=r.grad)
    hip, ref = res["hip"], res["torch"]

    def close(x, y, what, rtol=2e-5, atol=2e-6):
        scale = float(y.abs().max()) if y.numel() else 0.0
        err = float((x - y).abs().max()) if y.numel() else 0.0
        assert err <= atol + rtol * scale, f"{what}: {err:.3e} (scale {scale:.3e})"

    close(hip["rgb"], ref["rgb"], "rgb")
    if has_depth:
        close(hip["depth"], ref["depth"], "depth", 1e-6, 0)
    close(hip["f"][..., :10 if has_depth else 9], ref["f"][..., :10 if has_depth else 9], "grad feat", 1e-5, 1e-6)
    assert float(hip["f"][..., (10 if has_depth else 9):].abs().max() if CF > (10 if has_depth else 9) else 0.0) == 0.0
    close(hip["w1"], ref["w1"], "grad w1", 2e-5, 1e-5)
    close(hip["w2"], ref["w2"], "grad w2", 2e-5, 1e-5)
    if has_depth and depth_cot:
        close(hip["a"], ref["a"], "grad alphas", 1e-5, 1e-6)
    elif has_depth:
        assert hip["a"] is None or float(hip["a"].abs().max()) == 0.0
    if ray_map:
        close(hip["r"][0], ref["r"][0], "grad rays", 1e-5, 1e-6)
