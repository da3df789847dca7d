"""The BACKWARD half of the decoder fusion (round 6, mobgs_raster_bwd_decode): the Sandwich decoder's backward pass as the
prologue of the backward compositor against the separate decoder_bwd launch it replaces.  The compositing loop must see the
same cotangents bit for bit -- every splat-side gradient is compared with torch.equal -- while the decoder's weight and the
pose gradients, sums over all pixels taken in another order (per tile on the matrix pipe instead of per workgroup),
agree to rounding.  Covered: grids > 1024 tiles incl. the benchmark's own size (one wave per tile), heavy tiles inside the
quadrant kernel (four waves per tile: the partial rows meet in LDS), ragged image borders, a batch of cameras, gradient
sinks, and the fall-back when another output of the node carries a cotangent."""
import contextlib

import pytest
import torch

from test_gpu_fused_decode import _scene

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def _bwd_fusion(on, heavy_len=None):
    import mobgs_amd.rendering as R
    saved = (R.FUSE_DECODER_BWD, R.tuning.heavy_tile_len, R.tuning.bwd_mfma, R.path_log)
    R.FUSE_DECODER_BWD = on
    R.tuning.bwd_mfma = 0                       # the quadrant kernel whatever the grid
    if heavy_len is not None:
        R.tuning.heavy_tile_len = heavy_len
    R.path_log = []
    try:
        yield R
    finally:
        R.FUSE_DECODER_BWD, R.tuning.heavy_tile_len, R.tuning.bwd_mfma, R.path_log = saved


def _close(a, b, rtol=3e-4):
    scale = float(b.abs().max())
    return float((a - b).abs().max()) <= rtol * max(scale, 1e-20)


# 704x400 = 1100 tiles, the benchmark's 1352x1014 = 5440 tiles; 250x170 with heavy_len = 0: one wave per tile on a small
# grid with ragged borders (250 = 15 x 16 + 10, 170 = 10 x 16 + 10); heavy_len = 48 at 704x400: the longest eighth of the
# tiles are composited by four waves each INSIDE the quadrant kernel
@pytest.mark.parametrize("W,H,ns,nd,heavy_len", [(704, 400, 30_000, 15_000, None), (1352, 1014, 200_000, 100_000, None),
                                                  (250, 170, 4_000, 2_000, 0), (704, 400, 30_000, 15_000, 48),
                                                  (250, 170, 4_000, 2_000, 1)])
@pytest.mark.parametrize("with_depth", [True, False])
def test_render_with_the_decoder_prologue(hip_device, W, H, ns, nd, heavy_len, with_depth):
    from mobgs_amd.gaussian_renderer import render
    if (W, with_depth) == (1352, False):
        pytest.skip("the full size once")
    dev = hip_device
    v = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    vd = torch.randn(1, H, W, generator=torch.Generator().manual_seed(2)).to(dev)
    res, logs = {}, {}
    for fused in (False, True):
        with _bwd_fusion(fused, heavy_len) as R:
            cam, stat, dyn, _ = _scene(dev, W, H, ns, nd)
            cam.world_view_transform.requires_grad_(True)
            out = render(cam, stat, dyn, None, torch.zeros(9, device=dev))
            loss = (out["render"] * v).sum()
            if with_depth:
                loss = loss + (out["depth"] * vd).sum()
            loss.backward()
            res[fused] = ([out["render"].detach().clone(), out["depth"].detach().clone(), stat._xyz.grad.clone(),
                           stat._features_dc.grad.clone(), stat._scaling.grad.clone(), dyn.control_xyz.grad.clone(),
                           dyn._opacity.grad.clone(), dyn._features_dc.grad.clone(), dyn._features_t.grad.clone(),
                           out["viewspace_points"].grad.clone()],
                          [dyn.rgbdecoder.mlp1.weight.grad.clone(), dyn.rgbdecoder.mlp2.weight.grad.clone(),
                           cam.world_view_transform.grad.clone()])
            logs[fused] = [e for e in R.path_log if e["dir"] == "bwd" and e["D"] == 10][-1]
    assert logs[True].get("decode_bwd") and logs[True]["bwd_kernel"] == "quadrant" and not logs[False].get("decode_bwd")
    if heavy_len in (48, 1):
        assert logs[True]["heavy_tiles"] > 0
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[False][1], res[True][1]):
        assert torch.isfinite(b).all() and _close(b, a), (float((a - b).abs().max()), float(a.abs().max()))


def test_other_outputs_with_cotangents_take_the_separate_launch(hip_device):
    """A cotangent on the node's alpha output next to the decoded colour: the fused entry point adds it to the decoder's own
    alpha cotangent in registers -- same bits as the separate chain, which adds the two images in memory."""
    from mobgs_amd.gaussian_renderer import render
    dev = hip_device
    W, H = 704, 400
    v = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    res = {}
    for fused in (False, True):
        with _bwd_fusion(fused):
            cam, stat, dyn, _ = _scene(dev, W, H, 30_000, 15_000)
            out = render(cam, stat, dyn, None, torch.zeros(9, device=dev))
            alpha = out.get("alpha", None)
            loss = (out["render"] * v).sum() + out["depth"].sum()
            if alpha is not None:
                loss = loss + (alpha * alpha).sum()
            loss.backward()
            res[fused] = [stat._xyz.grad.clone(), dyn.control_xyz.grad.clone(), dyn._opacity.grad.clone(),
                          stat._features_dc.grad.clone()]
    for a, b in zip(res[False], res[True]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(a.abs().max()))


def test_render_many_with_the_decoder_prologue(hip_device):
    """A batch of cameras in one launch: one partial row per (camera, tile); the pose gradient of camera c sums its rows."""
    from mobgs_amd.camera import PinholeCamera
    from mobgs_amd.gaussian_renderer import render_many
    dev = hip_device
    W, H, K = 640, 480, 3     # 3 x 1200 tiles
    res = {}
    for fused in (False, True):
        with _bwd_fusion(fused, 0) as R:
            cam, stat, dyn, scam = _scene(dev, W, H, 20_000, 10_000)
            cams = []
            for k in range(K):
                pose = torch.eye(4)
                pose[0, 3] = 0.02 * k
                c = PinholeCamera(W, H, scam.K, pose, scam.time, scam.max_time, device=dev)
                c.world_view_transform.requires_grad_(True)
                cams.append(c)
            deltas = [torch.tensor(float(d), device=dev) for d in (-0.3, 0.0, 0.4)]
            outs = render_many(cams, stat, dyn, None, torch.zeros(9, device=dev), deltas)
            g = torch.Generator().manual_seed(3)
            loss = sum((o["render"] * torch.randn(3, H, W, generator=g).to(dev)).sum() + o["depth"].sum() for o in outs)
            loss.backward()
            res[fused] = ([o["render"].detach().clone() for o in outs] + [stat._xyz.grad.clone(), dyn.control_xyz.grad.clone(),
                                                                         dyn._features_t.grad.clone()],
                          [dyn.rgbdecoder.mlp1.weight.grad.clone(), dyn.rgbdecoder.mlp2.weight.grad.clone()] +
                          [c.world_view_transform.grad.clone() for c in cams])
            used = any(e.get("decode_bwd") for e in R.path_log)
            assert used == fused
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[False][1], res[True][1]):
        assert torch.isfinite(b).all() and _close(b, a), (float((a - b).abs().max()), float(a.abs().max()))


def test_gradient_sink_receives_the_decoder_gradients(hip_device):
    """ops.LeafGradSink: two renders back-propagated into the sink's buffers -- the decoder's weight gradients accumulate
    in-kernel (accumulate_wgrad) exactly as with the separate launch."""
    from mobgs_amd.gaussian_renderer import render
    from mobgs_amd.ops import LeafGradSink
    dev = hip_device
    W, H = 704, 400
    v = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    res = {}
    for fused in (False, True):
        with _bwd_fusion(fused):
            cam, stat, dyn, _ = _scene(dev, W, H, 30_000, 15_000)
            with LeafGradSink(stat, dyn):
                for rep in range(2):
                    out = render(cam, stat, dyn, None, torch.zeros(9, device=dev))
                    ((out["render"] * v).sum() * (rep + 1) + out["depth"].sum()).backward()
            res[fused] = [dyn.rgbdecoder.mlp1.weight.grad.clone(), dyn.rgbdecoder.mlp2.weight.grad.clone(),
                          stat._xyz.grad.clone(), dyn.control_xyz.grad.clone()]
    assert _close(res[True][0], res[False][0]) and _close(res[True][1], res[False][1])
    assert torch.equal(res[True][2], res[False][2]) and torch.equal(res[True][3], res[False][3])


def test_standalone_finish_equals_the_sums_inside_the_slot_reduction(hip_device):
    """mobgs_raster_bwd_decode_finish (a launch of its own, chunked + ticketed) against the same sums taken by the extra
    workgroups of the slot reduction (mobgs_raster_bwd_reduce_decode): equal to summation order; everything else bit for bit."""
    import mobgs_amd.rendering as R
    from mobgs_amd.gaussian_renderer import render
    dev = hip_device
    W, H = 704, 400
    v = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    res = {}
    saved = R.WGRAD_IN_REDUCE
    try:
        for inside in (True, False):
            R.WGRAD_IN_REDUCE = inside
            with _bwd_fusion(True):
                cam, stat, dyn, _ = _scene(dev, W, H, 30_000, 15_000)
                cam.world_view_transform.requires_grad_(True)
                out = render(cam, stat, dyn, None, torch.zeros(9, device=dev))
                ((out["render"] * v).sum() + out["depth"].sum()).backward()
                res[inside] = ([stat._xyz.grad.clone(), dyn.control_xyz.grad.clone(), dyn._features_t.grad.clone()],
                               [dyn.rgbdecoder.mlp1.weight.grad.clone(), dyn.rgbdecoder.mlp2.weight.grad.clone(),
                                cam.world_view_transform.grad.clone()])
    finally:
        R.WGRAD_IN_REDUCE = saved
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.isfinite(b).all() and _close(b, a)
