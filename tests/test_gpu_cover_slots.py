"""MobgsTuning.cover_slots (round 6): the backward compositor on the quadrant kernel writes EVERY gradient slot of its lists
itself -- zeros where no pixel blended the splat, the entries behind every pixel's last blended one included -- so the caller
no longer zero-fills the slot buffer (108 MB per 1352x1014 render).  Gradients must be bit-identical to the zero-filled path,
with the allocator's free blocks poisoned with NaNs before every backward pass (an unwritten slot would surface as NaN)."""
import pytest
import torch

from test_gpu_fused_decode import _scene

pytestmark = pytest.mark.gpu


def _poison(dev, mb=600):
    """Fill the caching allocator's free memory with NaNs: the next torch.empty of the slot buffer gets them."""
    blocks = [torch.full((mb * 1024 * 1024 // 4 // 4,), float("nan"), device=dev) for _ in range(4)]
    del blocks


@pytest.mark.parametrize("W,H,ns,nd,heavy_len,decode_bwd", [(704, 400, 30_000, 15_000, None, True),
                                                             (704, 400, 30_000, 15_000, 48, True),
                                                             (704, 400, 30_000, 15_000, 48, False),
                                                             (250, 170, 4_000, 2_000, 0, False),
                                                             (250, 170, 4_000, 2_000, 1, True),
                                                             (1352, 1014, 200_000, 100_000, None, True)])
def test_render_without_the_slot_fill_is_bit_identical(hip_device, W, H, ns, nd, heavy_len, decode_bwd):
    import mobgs_amd.rendering as R
    from mobgs_amd.gaussian_renderer import render
    dev = hip_device
    v = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    # a third of the image carries no cotangent at all: whole tiles whose walk never starts (top < first entry)
    v[:, : H // 3] = 0
    saved = (R.COVER_SLOTS, R.FUSE_DECODER_BWD, R.tuning.heavy_tile_len, R.tuning.bwd_mfma)
    res = {}
    try:
        R.tuning.bwd_mfma = 0
        R.FUSE_DECODER_BWD = decode_bwd
        if heavy_len is not None:
            R.tuning.heavy_tile_len = heavy_len
        for cover in (False, True):
            R.COVER_SLOTS = cover
            cam, stat, dyn, _ = _scene(dev, W, H, ns, nd)
            cam.world_view_transform.requires_grad_(True)
            out = render(cam, stat, dyn, None, torch.zeros(9, device=dev))
            loss = (out["render"] * v).sum()
            _poison(dev)
            loss.backward()
            res[cover] = [stat._xyz.grad.clone(), stat._features_dc.grad.clone(), stat._scaling.grad.clone(),
                          stat._opacity.grad.clone(), dyn.control_xyz.grad.clone(), dyn._opacity.grad.clone(),
                          dyn._features_t.grad.clone(), out["viewspace_points"].grad.clone(),
                          cam.world_view_transform.grad.clone(),
                          # (with the decoder prologue these two are sums over the per-tile rows of a torch.empty scratch:
                          # a row the kernel did not write would be one of the poison's NaNs)
                          dyn.rgbdecoder.mlp1.weight.grad.clone(), dyn.rgbdecoder.mlp2.weight.grad.clone()]
    finally:
        R.COVER_SLOTS, R.FUSE_DECODER_BWD, R.tuning.heavy_tile_len, R.tuning.bwd_mfma = saved
    for a, b in zip(res[False], res[True]):
        assert torch.isfinite(b).all()
        assert torch.equal(a, b)


def test_operator_level_backward_without_the_slot_fill(hip_device):
    """rasterization() (3 colour channels, two cameras, culling on) -- the plain operator path through the same node."""
    import mobgs_amd.rendering as R
    from mobgs_amd.rendering import rasterization
    from mobgs_amd.synth import SynthCamera, gaussian_cloud
    dev = hip_device
    W, H, N = 640, 480, 40_000
    scam = SynthCamera().scaled(W, H)
    cloud = gaussian_cloud(N, scam, 3)
    saved = (R.COVER_SLOTS, R.tuning.bwd_mfma)
    res = {}
    try:
        R.tuning.bwd_mfma = 0
        for cover in (False, True):
            R.COVER_SLOTS = cover
            means = cloud["xyz"].to(dev).requires_grad_(True)
            quats = torch.nn.functional.normalize(cloud["rotation"].to(dev), dim=-1).requires_grad_(True)
            scales = torch.exp(cloud["scaling"].to(dev)).requires_grad_(True)
            opac = torch.sigmoid(cloud["opacity"].to(dev)).reshape(-1).requires_grad_(True)
            cols = torch.rand(N, 3, generator=torch.Generator().manual_seed(5)).to(dev).requires_grad_(True)
            view = torch.eye(4, device=dev).repeat(2, 1, 1)
            view[1, 0, 3] = 0.1
            Ks = scam.K.to(dev).repeat(2, 1, 1)
            img, alpha, _ = rasterization(means, quats, scales, opac, cols, view, Ks, W, H, packed=False)
            g = torch.Generator().manual_seed(6)
            loss = (img * torch.randn(img.shape, generator=g).to(dev)).sum() + (alpha * alpha).sum()
            _poison(dev, 200)
            loss.backward()
            res[cover] = [t.grad.clone() for t in (means, quats, scales, opac, cols)]
    finally:
        R.COVER_SLOTS, R.tuning.bwd_mfma = saved
    for a, b in zip(res[False], res[True]):
        assert torch.isfinite(b).all() and torch.equal(a, b)
