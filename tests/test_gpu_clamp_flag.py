"""raster_bwd's clamp-free evaluation (round 6, csrc/raster.hip eval_splat_bwd): entries whose opacity cannot reach the
0.999 clamp skip the v_min / v_cmp / mask AND of every evaluated quadrant; entries that can take them behind a scalar
branch.  Opacities of 1.0 / 0.9995 / 0.99899 / 0.5 next to each other, dead-centre over pixels (where raw = opacity > 0.999:
the clamp IS active and alpha has zero slope): every gradient must agree with the C oracle, which evaluates upstream's
`if (opac * vis <= 0.999f)` per pair."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H", [(704, 400), (96, 64)])
def test_opacities_around_the_clamp_match_the_c_oracle(hip_device, W, H):
    from helpers import close
    from mobgs_amd import rendering
    from mobgs_amd.rendering import rasterization
    from mobgs_amd.synth import SynthCamera, splat_inputs
    from oracle import gsplat_cpu as Cc
    dev = hip_device
    N = 6000 if W > 200 else 400
    scam = SynthCamera().scaled(W, H)
    s = splat_inputs(N, scam, 11, 9)
    g = torch.Generator().manual_seed(12)
    # a third of the splats exactly / nearly opaque, big enough to own pixels outright
    pick = torch.randperm(N, generator=g)
    vals = torch.tensor([1.0, 0.9995, 0.99899, 0.999, 0.99905])
    s["opacities"][pick[: N // 3]] = vals[torch.randint(0, 5, (N // 3,), generator=g)]
    s["scales"][pick[: N // 6]] *= 3.0
    v = torch.randn(1, H, W, 10, generator=g)
    saved = (rendering.tuning.heavy_tile_len, rendering.tuning.bwd_mfma)
    rendering.tuning.heavy_tile_len, rendering.tuning.bwd_mfma = 0, 0   # the quadrant backward, one wave per tile
    try:
        t = {k: x.to(dev).clone().requires_grad_(k in ("means", "quats", "scales", "opacities", "colors"))
             for k, x in s.items()}
        img, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], t["viewmats"],
                                     t["Ks"], W, H, packed=False, backgrounds=torch.zeros(1, 9, device=dev),
                                     render_mode="RGB+D")
        (img * v.to(dev)).sum().backward()
    finally:
        rendering.tuning.heavy_tile_len, rendering.tuning.bwd_mfma = saved
    r = Cc.rasterization_fwd_bwd(*(s[k].numpy() for k in ["means", "quats", "scales", "opacities", "colors", "viewmats",
                                                          "Ks"]), W, H, backgrounds=np.zeros((1, 9), np.float32),
                                 render_mode="RGB+D", v_render=v.numpy())
    # the clamp must actually have been active somewhere: pixels whose alpha is exactly 0.999
    assert float(torch.from_numpy(r["alphas"]).max()) >= 0.999
    for k, ck in [("means", "v_means"), ("quats", "v_quats"), ("scales", "v_scales"), ("opacities", "v_opacities"),
                  ("colors", "v_colors")]:
        ref = torch.from_numpy(r[ck])
        sc = float(ref.abs().max())
        close(t[k].grad, ref, 1e-3, 1e-4 * sc, f"grad[{k}]", flip_frac=2e-4, flip_atol=5e-3 * sc)
    # a clamped pair passes NO gradient to the opacity: a fully opaque splat covering whole pixels still has one
    # through its unclamped fringe -- but never NaN / inf
    assert torch.isfinite(t["opacities"].grad).all()
