"""Degenerate inputs through the public operator API on the GPU: nothing to render, nothing visible, images smaller
than a tile, several cameras with heavy tiles.  The speculative (no host sync) and synchronous binning paths must
agree and nothing may fault."""
import pytest
import torch

from mobgs_amd.synth import SynthCamera, splat_inputs

pytestmark = pytest.mark.gpu


def _inputs(n, w, h, seed, dev, channels=3):
    cam = SynthCamera().scaled(w, h)
    return {k: v.to(dev) for k, v in splat_inputs(n, cam, seed, channels).items()}


def test_empty_scene_renders_background(hip_device):
    from mobgs_amd.rendering import rasterization
    s = _inputs(4, 40, 24, 0, hip_device)
    empty = {k: (v[:0] if k in ("means", "quats", "scales", "opacities", "colors") else v) for k, v in s.items()}
    bg = torch.tensor([[0.25, 0.5, 0.75]], device=hip_device)
    img, a, meta = rasterization(empty["means"], empty["quats"], empty["scales"], empty["opacities"],
                                 empty["colors"], empty["viewmats"], empty["Ks"], 40, 24, packed=False,
                                 backgrounds=bg)
    assert img.shape == (1, 24, 40, 3) and float(a.abs().max()) == 0.0
    assert torch.equal(img, bg.view(1, 1, 1, 3).expand_as(img))
    assert meta["flatten_ids"].numel() == 0


def test_all_splats_behind_the_camera(hip_device):
    from mobgs_amd import rendering
    s = _inputs(500, 64, 48, 1, hip_device)
    means = s["means"].clone()
    means[:, 2] = -means[:, 2].abs() - 1.0
    means.requires_grad_(True)
    for spec in (True, False):
        rendering.SPECULATIVE_BINNING = spec
        try:
            sp = rendering.SharedProjection(means, s["quats"], s["scales"], s["opacities"], s["viewmats"], s["Ks"],
                                            64, 48)
            img, a = sp.composite(torch.rand(500, 9, device=hip_device))
            assert float(img.abs().max()) == 0.0 and float(a.abs().max()) == 0.0
            assert int((sp.radii > 0).sum()) == 0 and sp.tl.n_isects == 0
            img.sum().backward()
            assert means.grad is None or float(means.grad.abs().max()) == 0.0
        finally:
            rendering.SPECULATIVE_BINNING = True


@pytest.mark.parametrize("w,h", [(8, 8), (17, 5), (16, 33)])
def test_images_smaller_than_or_straddling_a_tile(hip_device, w, h):
    """Speculative and synchronous binning agree bit for bit; pixels outside the image are never written."""
    from mobgs_amd import rendering
    s = _inputs(300, w, h, 2, hip_device, channels=9)
    out = {}
    for spec in (True, False):
        rendering.SPECULATIVE_BINNING = spec
        try:
            t = {k: v.clone().requires_grad_(k in ("means", "colors")) for k, v in s.items()}
            sp = rendering.SharedProjection(t["means"], t["quats"], t["scales"], t["opacities"], t["viewmats"],
                                            t["Ks"], w, h)
            img, a = sp.composite(t["colors"])
            assert img.shape == (1, h, w, 10) and bool(torch.isfinite(img).all())
            (img.sum() + a.sum()).backward()
            out[spec] = (img.detach().cpu(), t["means"].grad.cpu(), t["colors"].grad.cpu())
        finally:
            rendering.SPECULATIVE_BINNING = True
    for x, y in zip(out[True], out[False]):
        assert torch.equal(x, y)
    assert float(out[True][0].abs().max()) > 0.0


def test_two_cameras_with_heavy_tiles(hip_device):
    """C = 2 with a workgroup-per-tile schedule: equals rendering the cameras one by one."""
    from mobgs_amd import _lib, rendering
    lib = _lib.load()
    n, w, h = 3000, 96, 64
    s = _inputs(n, w, h, 3, hip_device, channels=9)
    s["scales"] = s["scales"] * 3.0
    vm2 = s["viewmats"].clone()
    vm2[0, 0, 3] += 0.15
    viewmats = torch.cat([s["viewmats"], vm2], 0)
    Ks = torch.cat([s["Ks"], s["Ks"]], 0)
    old = rendering.tuning.heavy_tile_len
    rendering.tuning.heavy_tile_len = 32
    try:
        both = rendering.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], viewmats, Ks,
                                       w, h, packed=False, render_mode="RGB+ED")
        singles = [rendering.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"],
                                           viewmats[c:c + 1], Ks[c:c + 1], w, h, packed=False, render_mode="RGB+ED")
                   for c in range(2)]
    finally:
        rendering.tuning.heavy_tile_len = old
    for c in range(2):
        assert torch.equal(both[0][c], singles[c][0][0]) and torch.equal(both[1][c], singles[c][1][0])


def test_4k_image_many_tiles(hip_device):
    """3840x2160 (32 400 tiles: more than the LDS-resident per-tile tables of some variants hold) with 200 k splats:
    speculative and synchronous binning agree bit for bit, forward + backward finite."""
    from mobgs_amd import rendering
    w, h = 3840, 2160
    s = _inputs(200_000, w, h, 9, hip_device, channels=9)
    out = {}
    for spec in (True, False):
        rendering.SPECULATIVE_BINNING = spec
        try:
            t = {k: v.clone().requires_grad_(k in ("means", "colors")) for k, v in s.items()}
            sp = rendering.SharedProjection(t["means"], t["quats"], t["scales"], t["opacities"], t["viewmats"],
                                            t["Ks"], w, h)
            img, a = sp.composite(t["colors"])
            (img.sum() + a.sum()).backward()
            assert bool(torch.isfinite(img).all()) and bool(torch.isfinite(t["means"].grad).all())
            out[spec] = (img.detach().cpu(), t["means"].grad.cpu(), sp.tl.n_isects)
        finally:
            rendering.SPECULATIVE_BINNING = True
    assert out[True][2] == out[False][2] and out[True][2] > 100_000
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
