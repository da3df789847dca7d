"""The oracle chain for one blurry training view (BLCE -> 9 latent cameras -> oracle render x9 -> mean) against
tests/golden/blurry_view.npz, which the reference's own scene.blce.BLCE and gaussian_renderer.render produced
(tests/golden/make_golden.py gen_blurry_view; train.py:441-541).  Runs on CPU."""
import numpy as np
import torch

from helpers import close, leaf_map, load, scene_from_fixture
from mobgs_amd.blce import BLCE, WarpedCamera, compute_frequency_blur_feature


def blce_from_fixture(fx, device="cpu"):
    idx, num_views = (int(v) for v in fx["in_idx"])
    m = BLCE(num_views=num_views, view_dim=32, num_warp=9)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("sd_")}, strict=True)
    return m.to(device), idx


def test_oracle_blurry_view_matches_reference_fixture():
    from oracle import render_torch as R
    fx = load("blurry_view")
    cam, stat, dyn, bg, w2c = scene_from_fixture(fx, "cpu")
    model, idx = blce_from_fixture(fx)
    blur = compute_frequency_blur_feature(torch.from_numpy(fx["in_image"]))
    warped_c2w, expo = model(torch.inverse(w2c), blur, idx)
    close(warped_c2w, fx["out_warped_c2w"], 1e-5, 1e-6, "warped c2w")
    close(expo, fx["out_exposure"], 1e-6, 1e-7, "exposure")
    warped_w2c = torch.inverse(warped_c2w)
    mid = R.render(cam, stat, dyn, bg, get_static=True, get_dynamic=True)
    frames = []
    for k in range(9):
        if k == 4:
            frames.append(mid["render"])
        else:
            frames.append(R.render(WarpedCamera(cam, warped_w2c[k], warped_c2w[k]), stat, dyn, bg, get_static=True,
                                   get_dynamic=True, delta_exposure=expo[k])["render"])
    pred = torch.stack(frames).mean(0) + 1e-10
    close(pred, fx["out_pred"], 1e-5, 1e-5, "blurry prediction")
    T = torch.from_numpy
    ((pred * T(fx["cot_v_pred"])).sum() + (mid["depth"] * T(fx["cot_v_depth"])).sum()
     + (mid["d_alpha"] * T(fx["cot_v_depth"])).sum()).backward()
    for k, leaf in leaf_map(stat, dyn).items():
        ref = fx["grad_" + k]
        close(leaf.grad, ref, 1e-4, 1e-5 * float(np.abs(ref).max()) + 1e-8, f"grad {k}")
    n = 0
    for k, p in model.named_parameters():
        if "bgrad_" + k in fx:
            ref = fx["bgrad_" + k]
            close(p.grad, ref, 1e-3, 1e-4 * float(np.abs(ref).max()), f"BLCE grad {k}")
            n += 1
    assert n >= 20
