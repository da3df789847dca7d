"""CPU tests of the ORACLE (-m "not gpu"): the restatements agree with each other, with known answers, and with
the golden fixtures the reference's own code produced (tests/golden/make_golden.py)."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import close, leaf_map, load, render_loss, scene_from_fixture
from mobgs_amd.synth import SynthCamera, splat_inputs
from oracle import gsplat_cpu as Cc
from oracle import gsplat_torch as G
from oracle import render_torch as R


# ------------------------------------------------------------------------------------------------------
# rasterizer restatements: C (upstream's kernels incl. hand backward) vs torch (independent, autograd)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode,channels,use_bg", [("RGB+ED", 9, True), ("RGB", 1, True), ("RGB", 2, False)])
def test_c_oracle_matches_torch_oracle(mode, channels, use_bg):
    n, w, h = 1200, 104, 72
    s = splat_inputs(n, SynthCamera().scaled(w, h), 4, channels)
    s["viewmats"] = s["viewmats"].clone()
    s["viewmats"][0, :3, 3] = torch.tensor([0.02, -0.01, 0.05])
    bg = torch.rand(1, channels, generator=torch.Generator().manual_seed(1)) if use_bg else None
    names = ["means", "quats", "scales", "opacities", "colors", "viewmats"]
    t = {k: v.clone().requires_grad_(k in names) for k, v in s.items()}
    img, a, meta = G.rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], t["viewmats"],
                                   t["Ks"], w, h, packed=False, backgrounds=bg, render_mode=mode)
    meta["means2d"].retain_grad()
    g = torch.Generator().manual_seed(7)
    v_img = torch.randn(img.shape, generator=g)
    v_a = torch.randn(a.shape, generator=g)
    ((img * v_img).sum() + (a * v_a).sum()).backward()
    r = Cc.rasterization_fwd_bwd(*(s[k].numpy() for k in ["means", "quats", "scales", "opacities", "colors",
                                                          "viewmats", "Ks"]), w, h,
                                 backgrounds=None if bg is None else bg.numpy(), render_mode=mode,
                                 v_render=v_img.numpy(), v_alphas=v_a[..., 0].numpy())
    assert np.array_equal(r["radii"], meta["radii"].numpy())
    assert np.array_equal(r["tiles_per_gauss"], meta["tiles_per_gauss"].numpy())
    assert np.array_equal(r["flatten_ids"], meta["flatten_ids"].numpy())
    assert np.array_equal(r["isect_ids"], meta["isect_ids"].numpy())
    assert np.array_equal(r["isect_offsets"], meta["isect_offsets"].numpy())
    scale = max(1.0, float(img.detach().abs().max()))
    close(r["render"], img, 0, 2e-5 * scale, "image", flip_frac=1e-3, flip_atol=scale / 255)
    close(r["alphas"], a[..., 0], 0, 2e-5, "alpha", flip_frac=1e-3, flip_atol=1 / 255)
    for k, ck in [("means", "v_means"), ("quats", "v_quats"), ("scales", "v_scales"), ("opacities", "v_opacities"),
                  ("colors", "v_colors"), ("viewmats", "v_viewmats")]:
        ref = t[k].grad
        close(r[ck], ref, 1e-3, 5e-4 * float(ref.abs().max()) + 1e-6, f"grad[{k}]")
    ref = meta["means2d"].grad
    close(r["v_means2d"], ref, 1e-3, 5e-4 * float(ref.abs().max()), "grad[means2d]")


# ------------------------------------------------------------------------------------------------------
# known answers (the reference has no tests of its own; SURVEY.md section 8c list)
# ------------------------------------------------------------------------------------------------------
def _one(means, scales, opac, W=64, H=48, f=50.0, quat=(1.0, 0, 0, 0), colors=None, bg=None):
    n = means.shape[0]
    K = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]])[None]
    q = torch.tensor([quat] * n, dtype=torch.float32)
    cols = torch.ones(n, 1) if colors is None else colors
    return G.rasterization(means, q, scales, opac, cols, torch.eye(4)[None], K, W, H, packed=False, backgrounds=bg)


def test_single_isotropic_gaussian_analytic():
    W, H, f, z, s, o = 64, 48, 50.0, 2.0, 0.1, 0.8
    img, a, meta = _one(torch.tensor([[0.0, 0.0, z]]), torch.full((1, 3), s), torch.tensor([o]), W, H, f)
    var = (f * s / z) ** 2 + 0.3
    assert int(meta["radii"][0, 0]) == math.ceil(3 * math.sqrt(var + math.sqrt(0.01)))  # b + sqrt(max(0.01, 0))
    assert torch.allclose(meta["means2d"][0, 0], torch.tensor([W / 2, H / 2]))
    assert torch.allclose(meta["conics"][0, 0], torch.tensor([1 / var, 0.0, 1 / var]), rtol=1e-6)
    # pixel (33, 24): centre at (33.5, 24.5) -> d = (-1.5, -0.5) from the mean (32, 24)
    d2 = 1.5 ** 2 + 0.5 ** 2
    exp_alpha = o * math.exp(-0.5 * d2 / var)
    assert abs(float(a[0, 24, 33, 0]) - exp_alpha) < 1e-6
    r = int(meta["radii"][0, 0])
    tiles = (math.ceil((32 + r) / 16) - math.floor((32 - r) / 16)) * (math.ceil((24 + r) / 16) - math.floor((24 - r) / 16))
    assert int(meta["tiles_per_gauss"][0, 0]) == tiles


def test_equal_depth_ties_resolve_by_index():
    means = torch.tensor([[0.0, 0.0, 2.0], [0.0, 0.0, 2.0]])
    cols = torch.tensor([[1.0, 0.0], [0.0, 1.0]])
    img, a, meta = _one(means, torch.full((2, 3), 0.2), torch.tensor([0.9, 0.9]), colors=cols)
    ids = meta["flatten_ids"].reshape(-1, 2)
    assert (ids[:, 0] == 0).all() and (ids[:, 1] == 1).all()
    # the first splat is composited in front: its channel dominates at the centre
    assert img[0, 24, 32, 0] > img[0, 24, 32, 1]


def test_culling_near_plane_and_image_border():
    means = torch.tensor([[0.0, 0.0, 0.005], [0.0, 0.0, -1.0], [50.0, 0.0, 2.0], [0.0, 0.0, 2.0]])
    img, a, meta = _one(means, torch.full((4, 3), 0.05), torch.full((4,), 0.5))
    assert meta["radii"][0].tolist()[:3] == [0, 0, 0] and int(meta["radii"][0, 3]) > 0


def test_opacity_one_is_clamped_to_0_999():
    img, a, _ = _one(torch.tensor([[0.0, 0.0, 2.0]]), torch.full((1, 3), 5.0), torch.tensor([1.0]))
    assert abs(float(a[0, 24, 32, 0]) - 0.999) < 1e-6  # o*exp(-sigma) = 0.99998 > 0.999 at the centre pixel


def test_long_list_early_stop_and_background():
    n = 400  # > 256 splats in one tile; opaque ones in front stop the pixel at T <= 1e-4
    g = torch.Generator().manual_seed(0)
    means = torch.cat([0.02 * torch.randn(n, 2, generator=g), 2.0 + torch.rand(n, 1, generator=g)], dim=1)
    cols = torch.rand(n, 3, generator=g)
    bg = torch.tensor([[0.3, 0.6, 0.9]])
    img, a, meta = _one(means, torch.full((n, 3), 0.1), torch.full((n,), 0.9), colors=cols, bg=bg)
    r = Cc.rasterization_fwd_bwd(means.numpy(), np.tile([1.0, 0, 0, 0], (n, 1)), np.full((n, 3), 0.1), np.full(n, 0.9),
                                 cols.numpy(), np.eye(4)[None], np.array([[[50.0, 0, 32], [0, 50.0, 24], [0, 0, 1]]]),
                                 64, 48, backgrounds=bg.numpy())
    close(r["render"], img, 0, 2e-5, "image", flip_frac=2e-3, flip_atol=1 / 255)
    centre_T = 1 - float(a[0, 24, 32, 0])
    assert centre_T <= 1e-3  # stopped: transmittance left just above the 1e-4 cut
    assert int(r["last_ids"][0, 24, 32]) < meta["flatten_ids"].numel() - 1


def test_backgrounds_none_leaves_uncovered_pixels_zero():
    img, a, _ = _one(torch.tensor([[0.0, 0.0, 2.0]]), torch.full((1, 3), 0.02), torch.tensor([0.9]))
    assert float(img[0, 0, 0, 0]) == 0.0 and float(a[0, 0, 0, 0]) == 0.0


def test_viewmat_gradient_finite_difference():
    n, w, h = 50, 48, 32
    s = splat_inputs(n, SynthCamera().scaled(w, h), 9, 3)
    vm = s["viewmats"].double().clone().requires_grad_(True)

    def f(v):
        img, a, _ = G.rasterization(s["means"].double(), s["quats"].double(), s["scales"].double(),
                                    s["opacities"].double(), s["colors"].double(), v, s["Ks"].double(), w, h,
                                    packed=False)
        return (img * img).sum()

    f(vm).backward()
    eps = 1e-6
    for (r, c) in [(0, 3), (1, 3), (2, 3), (0, 1), (2, 0)]:
        d = torch.zeros_like(vm)
        d[0, r, c] = eps
        fd = (f(vm.detach() + d) - f(vm.detach() - d)) / (2 * eps)
        assert abs(float(fd) - float(vm.grad[0, r, c])) <= 1e-3 * abs(float(fd)) + 1e-4


# ------------------------------------------------------------------------------------------------------
# the reference's own glue, pinned by fixtures it generated
# ------------------------------------------------------------------------------------------------------
def test_hermite_matches_reference_fixture():
    fx = load("hermite")
    ctrl = torch.from_numpy(fx["control"]).requires_grad_(True)
    ncp = torch.from_numpy(fx["ncp"])
    cot = torch.from_numpy(fx["cot"])
    for i, t in enumerate(fx["ts"]):
        out = R.hermite(ctrl, torch.tensor(float(t), dtype=torch.float32), ncp)
        close(out, fx["out"][i], 1e-6, 1e-6, f"hermite(t={t})")
        g, = torch.autograd.grad((out * cot).sum(), ctrl)
        close(g, fx["grad"][i], 1e-6, 1e-6, f"hermite grad(t={t})")


def test_sandwich_matches_reference_fixture():
    fx = load("sandwich")
    T = lambda k: torch.from_numpy(fx[k]).requires_grad_(True)  # noqa: E731
    feat, rays, w1, w2 = T("feat"), T("rays"), T("w1"), T("w2")
    out = R.sandwich(w1, w2, feat, rays)
    close(out, fx["out"], 1e-6, 1e-6, "sandwich")
    (out * torch.from_numpy(fx["cot"])).sum().backward()
    for k, t in (("feat", feat), ("rays", rays), ("w1", w1), ("w2", w2)):
        close(t.grad, fx["grad_" + k], 1e-5, 1e-6, f"sandwich grad {k}")


@pytest.mark.parametrize("name", ["render_lean", "render_train", "render_train_delta_flow"])
def test_render_restatement_matches_reference_fixture(name):
    fx = load(name)
    cam, stat, dyn, bg, w2c = scene_from_fixture(fx)
    get_static, get_dynamic, has_delta, delta, get_flow, use_w2c = fx["opt"]
    w2c_leaf = w2c.clone().requires_grad_(True) if use_w2c else None
    out = R.render(cam, stat, dyn, bg, get_static=bool(get_static), get_dynamic=bool(get_dynamic), w2c=w2c_leaf,
                   delta_exposure=torch.tensor(float(delta)) if has_delta else None, get_flow=bool(get_flow))
    for k in [k[4:] for k in fx if k.startswith("out_")]:
        ref = fx["out_" + k]
        if ref.dtype == np.bool_ or ref.dtype == np.int32:
            assert np.array_equal(out[k].numpy(), ref), k
        else:
            close(out[k], ref, 1e-5, 1e-5 * max(1.0, float(np.abs(ref).max())), f"out[{k}]")
    # keys the reference returns as None stay None
    for k in ("blending_factor", "world_coordinates", "splat_center", "labels", "centroids"):
        assert out[k] is None
    render_loss(out, fx).backward()
    for k, leaf in leaf_map(stat, dyn).items():
        if "grad_" + k in fx:
            ref = fx["grad_" + k]
            close(leaf.grad, ref, 1e-4, 1e-5 * float(np.abs(ref).max()) + 1e-7, f"grad[{k}]")
    if use_w2c:
        close(w2c_leaf.grad, fx["grad_w2c"], 1e-4, 1e-5 * float(np.abs(fx["grad_w2c"]).max()), "grad[w2c]")
    close(out["viewspace_points"].grad, fx["grad_viewspace_points"], 1e-4, 1e-6, "viewspace_points.grad")


def test_get_flow_restatement_matches_reference_fixture():
    fx = load("get_flow")
    cam, stat, dyn, bg, w2c = scene_from_fixture(fx, requires_grad=False)
    with torch.no_grad():
        e2m, m2e, img, alpha = R.get_flow(cam, stat, dyn, bg, torch.tensor(float(fx["opt"][0])))
    close(e2m, fx["out_exp2mid"], 1e-5, 1e-4, "exp2mid")
    close(m2e, fx["out_mid2exp"], 1e-5, 1e-4, "mid2exp")
    close(img, fx["out_latent_img"], 1e-5, 1e-5, "latent_img")
    close(alpha, fx["out_latent_alpha"], 1e-5, 1e-5, "latent_alpha")
    from mobgs_amd.camera import PinholeCamera
    cam_b = PinholeCamera(cam.image_width, cam.image_height, cam.K, torch.from_numpy(fx["in_w2c_b"]), cam.time,
                          cam.max_time)
    with torch.no_grad():
        f2d, fimg = R.get_flow_static(cam, cam_b, cam, stat)
    close(f2d, fx["out_static_flow_2d"], 1e-5, 1e-4, "static flow_2d")
    close(fimg, fx["out_static_flow_img"], 1e-5, 1e-4, "static flow image")


def _flow_grad_check(fx, get_flow_fn, get_flow_static_fn, device, rtol, atol_rel, flip=None):
    """shared by the CPU (oracle) and GPU (HIP) tests of tests/golden/get_flow_grad.npz"""
    from helpers import leaf_map
    from mobgs_amd.camera import PinholeCamera
    flip = flip or {}
    cam, stat, dyn, bg, w2c = scene_from_fixture(fx, device=device, requires_grad=True)
    T = lambda k: torch.from_numpy(fx[k]).to(device)  # noqa: E731
    outs = get_flow_fn(cam, stat, dyn, bg, torch.tensor(float(fx["opt"][0]), device=device))
    from helpers import flow_flip_bound
    derived = flip.get("flip_atol") == "derived"  # one-blend-step bounds instead of a flat allowance
    for got, key in zip(outs, ("out_exp2mid", "out_mid2exp", "out_latent_img", "out_latent_alpha")):
        fl = dict(flip)
        if derived:
            fl["flip_atol"] = flow_flip_bound(fx[key]) if key in ("out_exp2mid", "out_mid2exp") else 2.0 / 255.0
        close(got, fx[key], 1e-5, 2e-4, key, **fl)
    torch.autograd.backward(list(outs), [T(k) for k in ("cot_exp2mid", "cot_mid2exp", "cot_latent_img",
                                                         "cot_latent_alpha")])
    leaves = leaf_map(stat, dyn)
    n = 0
    for k, leaf in leaves.items():
        if "grad_" + k in fx:
            ref = fx["grad_" + k]
            sc = float(np.abs(ref).max())
            fl = {"flip_frac": flip["flip_frac"] * 5, "flip_atol": (0.01 if derived else 0.05) * sc} if flip else {}
            close(leaf.grad, ref, rtol, atol_rel * sc + 1e-9, f"get_flow grad {k}", **fl)
            n += 1
        else:
            assert leaf.grad is None or float(leaf.grad.abs().max()) == 0.0, k
        leaf.grad = None
    assert n >= 14
    cam_b = PinholeCamera(cam.image_width, cam.image_height, cam.K, T("in_w2c_b"), cam.time, cam.max_time,
                          device=device)
    f2d, fimg = get_flow_static_fn(cam, cam_b, cam, stat, dyn, bg)
    close(f2d, fx["out_static_flow_2d"], 1e-5, 2e-4, "static flow_2d")
    fl = dict(flip)
    if derived:
        fl["flip_atol"] = 2.0 * (1.001 / 255.0) * 2.0 * float(np.abs(fx["out_static_flow_2d"]).max())
    close(fimg, fx["out_static_flow_img"], 1e-5, 2e-4, "static flow image", **fl)
    torch.autograd.backward([f2d, fimg], [T("cot_static_flow_2d"), T("cot_static_flow_img")])
    for k in ("s__xyz", "s__scaling", "s__rotation", "s__opacity"):
        ref = fx["sgrad_" + k]
        sc = float(np.abs(ref).max())
        fl = {"flip_frac": flip["flip_frac"] * 5, "flip_atol": (0.01 if derived else 0.05) * sc} if flip else {}
        close(leaves[k].grad, ref, rtol, atol_rel * sc + 1e-9, f"get_flow_static grad {k}", **fl)


def test_get_flow_gradients_of_the_restatement_match_reference_fixture():
    """get_flow()'s backward was never compared with the reference (the old fixture is no_grad)."""
    _flow_grad_check(load("get_flow_grad"), lambda cam, s, d, bg, dl: R.get_flow(cam, s, d, bg, dl),
                     lambda a, b, c, s, d, bg: R.get_flow_static(a, b, c, s), "cpu", 1e-4, 1e-5)


@pytest.mark.skipif(not os.path.isdir("/root/reference/gaussian_renderer"), reason="reference tree not present")
def test_render_restatement_matches_reference_live():
    """When /root/reference is present (build container), run its render() directly on a fresh scene."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import ref_harness as RH
    from make_golden import ref_models, scene_params, small_w2c
    from mobgs_amd.camera import PinholeCamera
    gr = RH.ref_import("gaussian_renderer")
    W, H = 72, 56
    scam = SynthCamera().scaled(W, H)
    cam = PinholeCamera(W, H, scam.K, small_w2c(), time=0.3, max_time=scam.max_time)
    stat_p, dyn_p = scene_params(500, 300, scam, 21)
    spc, dpc = ref_models(stat_p, dyn_p, 21)
    bg = torch.zeros(9)
    with RH.CudaToCpu(), torch.no_grad():
        ref = gr.render(cam, spc, dpc, None, bg, get_static=True, get_dynamic=True,
                        delta_exposure=torch.tensor(-0.2), get_flow=True)
        mine = R.render(cam, spc, dpc, bg, get_static=True, get_dynamic=True, delta_exposure=torch.tensor(-0.2),
                        get_flow=True)
    assert set(ref.keys()) == set(mine.keys())
    for k, v in ref.items():
        if isinstance(v, torch.Tensor):
            if v.dtype in (torch.bool, torch.int32):
                assert torch.equal(v, mine[k]), k
            else:
                close(mine[k], v, 1e-5, 1e-5 * max(1.0, float(v.abs().max())), k)
        else:
            assert mine[k] is None, k


def test_deform_restatement_matches_reference_fixture():
    from oracle import deform_torch as D
    fx = load("deform")
    T = torch.from_numpy
    planes = [[T(fx[f"plane_{l}_{p}"]).requires_grad_(True) for p in range(6)] for l in range(3)]
    W = {k[2:]: T(v).requires_grad_(True) for k, v in fx.items() if k.startswith("w_")}
    pts, scales, rots = (T(fx[k]).requires_grad_(True) for k in ("in_pts", "in_scales", "in_rots"))
    o = D.deform_forward(pts, scales, rots, T(fx["in_times"]), T(fx["in_aabb"]), planes, W)
    for a, k in zip(o, ("out_pts", "out_scales", "out_rots")):
        close(a, fx[k], 1e-6, 1e-6, k)
    ((o[0] * T(fx["cot_pts"])).sum() + (o[1] * T(fx["cot_scales"])).sum() + (o[2] * T(fx["cot_rots"])).sum()).backward()
    close(pts.grad, fx["grad_pts"], 1e-5, 1e-6, "grad pts")
    close(W["w0"].grad, fx["gw_w0"], 1e-5, 1e-6, "grad w0")
    close(planes[2][0].grad, fx["gplane_2_0"], 1e-5, 1e-7, "grad plane 2.0")


def test_deform_restatement_matches_reference_at_mid_size_planes():
    """Second pin of oracle/deform_torch.py: planes [32,32,32,12] x [1,2,4] (tests/golden/deform_mid.npz, produced by
    the reference's deform_network), 4000 points, mixed per-call / per-point time stamps, points outside the box."""
    from helpers import check_deform_fixture_grads, mid_planes
    from oracle import deform_torch as D
    fx = load("deform_mid")
    T = torch.from_numpy
    seed, base, _ = (int(v) for v in fx["meta"])
    shapes = [[1, 32, (base * m if b < 3 else 12), (base * m if a < 3 else 12)] for m in (1, 2, 4) for a, b in D.COMBS]
    vals = mid_planes(shapes, seed)
    planes = [[vals[6 * l + p].clone().requires_grad_(True) for p in range(6)] for l in range(3)]
    W = {k[2:]: T(v).requires_grad_(True) for k, v in fx.items() if k.startswith("w_")}
    pts, scales, rots = (T(fx[k]).requires_grad_(True) for k in ("in_pts", "in_scales", "in_rots"))
    o = D.deform_forward(pts, scales, rots, T(fx["in_times"]), T(fx["in_aabb"]), planes, W)
    for a, k in zip(o, ("out_pts", "out_scales", "out_rots")):
        close(a, fx[k], 1e-6, 1e-6, k)
    ((o[0] * T(fx["cot_pts"])).sum() + (o[1] * T(fx["cot_scales"])).sum() + (o[2] * T(fx["cot_rots"])).sum()).backward()
    for n, t in (("pts", pts), ("scales", scales), ("rots", rots)):
        close(t.grad, fx["grad_" + n], 1e-5, 1e-6 * float(np.abs(fx["grad_" + n]).max()), "grad " + n)
    for k, w in W.items():
        close(w.grad, fx["gw_" + k], 1e-4, 1e-5 * float(np.abs(fx["gw_" + k]).max()), "grad " + k)
    check_deform_fixture_grads(fx, lambda li, pi: planes[li][pi].grad, "oracle ")


def test_loss_restatement_matches_reference_fixture():
    from oracle import loss_torch as L
    fx = load("losses")
    gt = torch.from_numpy(fx["gt"])
    img = torch.from_numpy(fx["img"]).requires_grad_(True)
    close(L.l1_loss(img, gt), fx["l1"], 1e-6, 1e-7, "l1")
    close(L.ssim(img, gt), fx["ssim"], 1e-6, 1e-7, "ssim")
    close(L.ssim(img, gt, size_average=False), fx["ssim_per_image"], 1e-6, 1e-7, "ssim per image")
    close(L.psnr(img.detach(), gt), fx["psnr"], 1e-6, 1e-6, "psnr")
    (L.l1_loss(img, gt) + 0.2 * (1 - L.ssim(img, gt))).backward()
    close(img.grad, fx["grad_img"], 1e-5, 1e-8, "grad")


def _assert_state(st, fx, tag, atol=0.0):
    from oracle import densify_torch as D
    for g in D.GROUPS:
        for kind, store in (("", st["params"]), (".exp_avg", st["exp_avg"]), (".exp_avg_sq", st["exp_avg_sq"])):
            key = f"{tag}.{g}{kind}"
            if key not in fx:
                assert kind and g not in store
                continue
            ref = torch.from_numpy(fx[key])
            got = store[g]
            assert got.shape == ref.shape, (key, got.shape, ref.shape)
            if atol == 0.0:
                assert torch.equal(got, ref), key
            else:
                assert torch.allclose(got, ref, rtol=1e-6, atol=atol), key
    for a in D.AUX:
        ref = torch.from_numpy(fx[f"{tag}.{a}"])
        assert st["aux"][a].shape == ref.shape and torch.equal(st["aux"][a], ref), f"{tag}.{a}"


def test_densify_oracle_matches_reference_fixture():
    """oracle/densify_torch.py replays the reference GaussianModel's statistics -> clone -> split -> prune ->
    opacity-reset sequence state for state (parameters, Adam moments, per-splat statistics)."""
    from oracle import densify_torch as D
    fx = load("densify")
    st = D.state_from_fixture(fx, "s0")
    for it in range(2):
        D.add_densification_stats(st, torch.from_numpy(fx[f"stats{it}.viewspace_grad"]),
                                  torch.from_numpy(fx[f"stats{it}.visible"]),
                                  torch.from_numpy(fx[f"stats{it}.radii"]))
    _assert_state(st, fx, "s1")
    grads = D.mean_grads(st)
    assert torch.equal(grads, torch.from_numpy(fx["grads"]))
    thr, extent = float(fx["max_grad"]), float(fx["extent"])
    D.densify_and_clone(st, grads, thr, extent)
    _assert_state(st, fx, "s2")
    D.densify_and_splitv2(st, grads, thr, extent, 2, samples=torch.from_numpy(fx["split.samples"]))
    _assert_state(st, fx, "s3")
    D.prune_points(st, torch.from_numpy(fx["prune.mask"]))
    _assert_state(st, fx, "s4")
    D.reset_opacity(st)
    _assert_state(st, fx, "s5")


def test_normals_oracle_matches_reference_fixture():
    from oracle import normals_torch as NT
    fx = load("normals")
    z = torch.from_numpy(fx["z"]).requires_grad_(True)
    k = [float(v) for v in fx["intrinsics"]]
    n = NT.get_normals(z + 1e-6, *k)
    assert torch.allclose(n, torch.from_numpy(fx["normals"]), rtol=0, atol=2e-6)
    (n * torch.from_numpy(fx["cotangent"])).sum().backward()
    ref = torch.from_numpy(fx["grad_z"])
    assert torch.allclose(z.grad, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))
