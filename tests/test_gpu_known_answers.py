"""SURVEY.md section 8c's known-answer list run against the HIP rasterizer itself (round 1 only ran it against the
oracle): analytic truths that need no second implementation -- closed-form alpha / radius / conic / tile rectangle of
one isotropic Gaussian, tie order, near-plane and border culls, the 0.999 alpha clamp, a 400-deep pixel composited by
hand in float64 (1/255 skip, 0.999 clamp, 1e-4 stop), background handling.  Semantics: gsplat 1.4.0 as restated in
SURVEY.md Appendix A (the reference calls it at gaussian_renderer/__init__.py:143-176)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _one(dev, means, scales, opac, W=64, H=48, f=50.0, quat=(1.0, 0, 0, 0), colors=None, bg=None, cull=True):
    from mobgs_amd import rendering
    n = means.shape[0]
    K = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]])[None].to(dev)
    q = torch.tensor([quat] * n, dtype=torch.float32).to(dev)
    cols = torch.ones(n, 1) if colors is None else colors
    rendering.set_tile_culling(cull)
    try:
        return rendering.rasterization(means.to(dev), q, scales.to(dev), opac.to(dev), cols.to(dev),
                                       torch.eye(4)[None].to(dev), K, W, H, packed=False,
                                       backgrounds=None if bg is None else bg.to(dev))
    finally:
        rendering.set_tile_culling(True)


def test_single_isotropic_gaussian_analytic(hip_device):
    W, H, f, z, s, o = 64, 48, 50.0, 2.0, 0.1, 0.8
    img, a, meta = _one(hip_device, torch.tensor([[0.0, 0.0, z]]), torch.full((1, 3), s), torch.tensor([o]), W, H, f,
                        cull=False)
    var = (f * s / z) ** 2 + 0.3
    r = int(meta["radii"][0, 0])
    assert r == math.ceil(3 * math.sqrt(var + math.sqrt(0.01)))  # b + sqrt(max(0.01, b^2 - det)), b^2 = det here
    assert torch.allclose(meta["means2d"][0, 0].cpu(), torch.tensor([W / 2, H / 2]))
    assert torch.allclose(meta["conics"][0, 0].cpu(), torch.tensor([1 / var, 0.0, 1 / var]), rtol=1e-6)
    assert abs(float(meta["depths"][0, 0]) - z) < 1e-6
    # every pixel: alpha = o exp(-d^2 / 2 var) where >= 1/255, else 0 (closed form, float64)
    ys, xs = np.mgrid[0:H, 0:W]
    d2 = (xs + 0.5 - W / 2) ** 2 + (ys + 0.5 - H / 2) ** 2
    alpha = o * np.exp(-0.5 * d2 / var)
    alpha = np.where(alpha >= 1 / 255, np.minimum(alpha, 0.999), 0.0)
    got = a[0, ..., 0].cpu().numpy().astype(np.float64)
    near_cut = np.abs(o * np.exp(-0.5 * d2 / var) - 1 / 255) < 1e-6  # pixels ON the threshold may fall either way
    assert np.abs(got - alpha)[~near_cut].max() < 2e-6
    assert np.abs(img[0, ..., 0].cpu().numpy() - alpha)[~near_cut].max() < 2e-6  # colour 1, no background
    tiles = (math.ceil((32 + r) / 16) - math.floor((32 - r) / 16)) * (math.ceil((24 + r) / 16) - math.floor((24 - r) / 16))
    assert int(meta["tiles_per_gauss"][0, 0]) == tiles
    assert meta["flatten_ids"].numel() == tiles  # culling off: gsplat's bounding-box list


def test_equal_depth_ties_resolve_by_index(hip_device):
    means = torch.tensor([[0.0, 0.0, 2.0], [0.0, 0.0, 2.0]])
    cols = torch.tensor([[1.0, 0.0], [0.0, 1.0]])
    img, a, meta = _one(hip_device, means, torch.full((2, 3), 0.2), torch.tensor([0.9, 0.9]), colors=cols, cull=False)
    ids = meta["flatten_ids"].reshape(-1, 2).cpu()
    assert (ids[:, 0] == 0).all() and (ids[:, 1] == 1).all()
    # the first splat is composited in front: centre pixel = (a, (1 - a) a) with a = 0.9 exp(-sigma)
    var = (50.0 * 0.2 / 2.0) ** 2 + 0.3
    al = 0.9 * math.exp(-0.5 * 0.5 / var)
    px = img[0, 24, 32].cpu().double()
    assert abs(float(px[0]) - al) < 2e-6 and abs(float(px[1]) - (1 - al) * al) < 2e-6


def test_culling_near_plane_and_image_border(hip_device):
    means = torch.tensor([[0.0, 0.0, 0.005], [0.0, 0.0, -1.0], [50.0, 0.0, 2.0], [0.0, 0.0, 2.0]])
    img, a, meta = _one(hip_device, means, torch.full((4, 3), 0.05), torch.full((4,), 0.5))
    assert meta["radii"][0].tolist()[:3] == [0, 0, 0] and int(meta["radii"][0, 3]) > 0
    # a splat straddling the right border stays, and only its on-screen tiles are listed
    img, a, meta = _one(hip_device, torch.tensor([[1.26, 0.0, 2.0]]), torch.full((1, 3), 0.1), torch.tensor([0.9]),
                        cull=False)  # projects to x = 63.5
    r = int(meta["radii"][0, 0])
    assert r > 0 and int(meta["tiles_per_gauss"][0, 0]) == (4 - math.floor((63.5 - r) / 16)) * \
        (math.ceil((24 + r) / 16) - math.floor((24 - r) / 16))
    assert float(a[0, 24, 63, 0]) > 0.5 and float(a[0, 24, 0, 0]) == 0.0


def test_opacity_one_is_clamped_to_0_999(hip_device):
    img, a, _ = _one(hip_device, torch.tensor([[0.0, 0.0, 2.0]]), torch.full((1, 3), 5.0), torch.tensor([1.0]))
    assert abs(float(a[0, 24, 32, 0]) - 0.999) < 1e-6  # o exp(-sigma) = 0.99998 > 0.999 at the centre pixel


def _compose_pixel(px, py, means2d, conics, opac, cols, order, bg):
    """gsplat's per-pixel loop in float64 from the projected splats (Appendix A.3)."""
    T, out = 1.0, np.zeros(cols.shape[1])
    last = -1
    for rank, i in enumerate(order):
        dx, dy = means2d[i, 0] - px, means2d[i, 1] - py
        sigma = 0.5 * (conics[i, 0] * dx * dx + conics[i, 2] * dy * dy) + conics[i, 1] * dx * dy
        alpha = min(0.999, opac[i] * math.exp(-sigma))
        if sigma < 0 or alpha < 1 / 255:
            continue
        nT = T * (1 - alpha)
        if nT <= 1e-4:
            break
        out += cols[i] * alpha * T
        T = nT
        last = rank
    return out + T * bg, 1 - T, last


def test_long_list_early_stop_and_background(hip_device):
    n = 400  # > 256 splats in one tile; opaque ones in front stop the pixel at T <= 1e-4
    g = torch.Generator().manual_seed(0)
    means = torch.cat([0.02 * torch.randn(n, 2, generator=g), 2.0 + torch.rand(n, 1, generator=g)], dim=1)
    cols = torch.rand(n, 3, generator=g)
    bg = torch.tensor([[0.3, 0.6, 0.9]])
    img, a, meta = _one(hip_device, means, torch.full((n, 3), 0.1), torch.full((n,), 0.9), colors=cols, bg=bg)
    m2d = meta["means2d"][0].cpu().double().numpy()
    con = meta["conics"][0].cpu().double().numpy()
    depth = meta["depths"][0].cpu().numpy()
    order = np.lexsort((np.arange(n), depth))  # depth, ties by index
    for (px, py) in [(32, 24), (30, 22), (36, 27), (20, 24)]:
        ref, ref_a, last = _compose_pixel(px + 0.5, py + 0.5, m2d, con, np.full(n, 0.9), cols.double().numpy(), order,
                                          bg[0].double().numpy())
        assert np.abs(img[0, py, px].cpu().double().numpy() - ref).max() < 5e-6, (px, py)
        assert abs(float(a[0, py, px, 0]) - ref_a) < 5e-6
    _, centre_a, last = _compose_pixel(32.5, 24.5, m2d, con, np.full(n, 0.9), cols.double().numpy(), order,
                                       bg[0].double().numpy())
    assert 1 - centre_a <= 1e-3 and last < n - 1, "the centre pixel must stop early on the 1e-4 transmittance rule"
    # an uncovered corner shows the background
    assert torch.allclose(img[0, 0, 0].cpu(), bg[0], atol=1e-7) and float(a[0, 0, 0, 0]) == 0.0


def test_backgrounds_none_leaves_uncovered_pixels_zero(hip_device):
    img, a, _ = _one(hip_device, torch.tensor([[0.0, 0.0, 2.0]]), torch.full((1, 3), 0.02), torch.tensor([0.9]))
    assert float(img[0, 0, 0, 0]) == 0.0 and float(a[0, 0, 0, 0]) == 0.0
    assert float(a[0, 24, 32, 0]) > 0.0


def test_single_gaussian_gradients_closed_form(hip_device):
    """d(sum of image)/d(opacity) and d/d(colour) of one Gaussian without overlap: sum_p alpha_p / o and sum_p alpha_p."""
    from mobgs_amd import rendering
    dev = hip_device
    W, H, f, z, s, o = 64, 48, 50.0, 2.0, 0.1, 0.6
    K = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]])[None].to(dev)
    opac = torch.tensor([o], device=dev, requires_grad=True)
    col = torch.tensor([[0.7]], device=dev, requires_grad=True)
    img, a, _ = rendering.rasterization(torch.tensor([[0.0, 0.0, z]], device=dev),
                                        torch.tensor([[1.0, 0, 0, 0]], device=dev), torch.full((1, 3), s, device=dev),
                                        opac, col, torch.eye(4)[None].to(dev), K, W, H, packed=False)
    img.sum().backward()
    var = (f * s / z) ** 2 + 0.3
    ys, xs = np.mgrid[0:H, 0:W]
    d2 = (xs + 0.5 - W / 2) ** 2 + (ys + 0.5 - H / 2) ** 2
    alpha = o * np.exp(-0.5 * d2 / var)
    alpha = np.where(alpha >= 1 / 255, alpha, 0.0)
    assert abs(float(col.grad) - alpha.sum()) < 1e-4 * alpha.sum()
    assert abs(float(opac.grad) - 0.7 * alpha.sum() / o) < 1e-4 * alpha.sum()
