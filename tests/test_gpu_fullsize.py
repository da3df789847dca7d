"""Full-size checks on the GPU (BASELINE config #2: 300k splats, 1352x1014; config #5 scale: 800k splats).

Direct comparison with the C oracle at full size (it finishes in seconds on the GPU box's host cores) plus
size-independent properties: sorted lists, linearity in the colours, culling on/off identity, fp16 attribute storage.
"""
import numpy as np
import pytest
import torch

from helpers import close, close_image_with_blend_flips, psnr
from mobgs_amd.synth import SynthCamera, splat_inputs

pytestmark = pytest.mark.gpu
W, H = 1352, 1014


@pytest.fixture(scope="module")
def scene300k():
    cam = SynthCamera()
    return splat_inputs(300_000, cam, 0, 9), cam


def _run(s, dev, mode="RGB+ED", bg=None, grads=False, colors=None):
    from mobgs_amd.rendering import rasterization
    names = ["means", "quats", "scales", "opacities", "colors", "viewmats"]
    t = {k: v.to(dev).clone().requires_grad_(grads and k in names) for k, v in s.items()}
    if colors is not None:
        t["colors"] = colors.to(dev)
    img, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], t["viewmats"],
                                 t["Ks"], W, H, packed=False, backgrounds=bg, render_mode=mode)
    return t, img, a, meta


def test_fullsize_forward_backward_against_c_oracle(hip_device, scene300k):
    from oracle import gsplat_cpu as Cc
    s, _ = scene300k
    g = torch.Generator().manual_seed(100)
    v_img = torch.randn(1, H, W, 10, generator=g)
    t, img, a, meta = _run(s, hip_device, grads=True, bg=torch.zeros(1, 9, device=hip_device))
    meta["means2d"].retain_grad()
    (img * v_img.to(hip_device)).sum().backward()
    r = Cc.rasterization_fwd_bwd(*(s[k].numpy() for k in ["means", "quats", "scales", "opacities", "colors",
                                                          "viewmats", "Ks"]), W, H,
                                 backgrounds=np.zeros((1, 9), np.float32), render_mode="RGB+ED",
                                 v_render=v_img.numpy())
    # a handful of splats out of 300 000 sit within an ulp of a discrete decision (3-sigma radius ceil, image-border
    # cull) and fall on the other side when the projection is contracted into FMAs differently
    rad = meta["radii"].cpu().numpy().astype(np.int64)
    ref_rad = r["radii"].astype(np.int64)
    differ = rad != ref_rad
    # observed: 0 of 300 000 since project.hip evaluates the expressions as written (-ffp-contract=off, round 2); a
    # regression of the kind the soak found (1 splat in 8000) would show as ~40
    print(f"\n[fullsize] radii differing from the C oracle: {int(differ.sum())} of {rad.size}")
    assert differ.sum() == 0, int(differ.sum())
    both = differ & (rad > 0) & (ref_rad > 0)
    assert np.abs(rad - ref_rad)[both].max(initial=0) <= 1
    ref = torch.from_numpy(r["render"])
    scale = float(ref.abs().max())
    # SURVEY's criterion (max |dpixel| <= 2e-5 of the range) for every pixel except those where an alpha-threshold
    # decision flips; those are bounded by the flipped splat's own weight (helpers.close_image_with_blend_flips)
    vis_depth = meta["depths"][meta["radii"] > 0]
    nflip, worst = close_image_with_blend_flips(img, ref, r["alphas"], float(s["colors"].abs().max()),
                                                float(vis_depth.max() - vis_depth.min()), 3e-5 * scale, "image",
                                                flip_frac=2e-4, n_colour_channels=9, alphas_img=a)  # observed 2e-6 .. 6e-5
    print(f"\n[fullsize] image: {nflip} of {ref.numel()} elements beyond 3e-5 x range (largest {worst:.2e}); all "
          "within the one-blend-step bound")
    target = (ref[..., :9] / scale + 0.05 * torch.randn(ref[..., :9].shape, generator=g))
    assert abs(psnr(img[..., :9].cpu() / scale, target) - psnr(ref[..., :9] / scale, target)) <= 1e-4
    for k, ck in [("means", "v_means"), ("quats", "v_quats"), ("scales", "v_scales"), ("opacities", "v_opacities"),
                  ("colors", "v_colors"), ("viewmats", "v_viewmats")]:
        refg = torch.from_numpy(r[ck])
        sc = float(refg.abs().max())
        # observed: no entry beyond rtol 1e-3 + 1e-4 max, largest error 9e-5 of the maximum.  A flipped blend decision
        # moves the gradients of the splats of ONE pixel by at most that pixel's share: allowed for 1e-5 of the entries
        # (3 - 27 of them) up to 5e-3 of the maximum
        close(t[k].grad, refg, 1e-3, 1e-4 * sc, f"grad[{k}]", flip_frac=1e-5, flip_atol=5e-3 * sc)


def test_fullsize_lists_are_sorted_and_complete(hip_device, scene300k):
    s, _ = scene300k
    _, _, _, meta = _run(s, hip_device, mode="RGB")
    ids = meta["isect_ids"]
    assert bool((ids[1:] >= ids[:-1]).all()), "isect_ids (tile | depth bits) must be non-decreasing"
    flat = meta["flatten_ids"].long()
    depth_bits = meta["depths"].reshape(-1)[flat].view(torch.int32).long()
    assert torch.equal(ids & 0xFFFFFFFF, depth_bits)
    same = ids[1:] == ids[:-1]
    assert bool((flat[1:][same] > flat[:-1][same]).all()), "equal keys must be ordered by splat index"
    offs = meta["isect_offsets"].reshape(-1).long()
    tile = ids >> 32
    assert torch.equal(torch.searchsorted(tile.contiguous(), torch.arange(offs.numel(), device=ids.device)), offs)


def test_fullsize_linearity_in_colours_and_culling_identity(hip_device, scene300k):
    from mobgs_amd import rendering
    s, _ = scene300k
    g = torch.Generator().manual_seed(5)
    c1, c2 = torch.randn(300_000, 3, generator=g), torch.randn(300_000, 3, generator=g)
    imgs = {}
    for name, c in (("a", c1), ("b", c2), ("ab", c1 + c2)):
        imgs[name] = _run(s, hip_device, mode="RGB", colors=c)[1]
    close(imgs["ab"], imgs["a"] + imgs["b"], 1e-5, 2e-5 * float(imgs["ab"].abs().max()), "linearity")
    rendering.set_tile_culling(False)
    try:
        full = _run(s, hip_device, mode="RGB", colors=c1)[1]
    finally:
        rendering.set_tile_culling(True)
    assert torch.equal(full, imgs["a"]), "reach culling changed pixels"


def test_fp16_attribute_storage(hip_device, scene300k):
    """BASELINE config #5 stores Gaussian attributes in fp16 (no such mode exists in the reference): inputs of any
    float dtype are widened on load, compute stays fp32.  Reported, not a parity bar: PSNR vs the fp32 render."""
    s, _ = scene300k
    ref = _run(s, hip_device, mode="RGB", colors=torch.sigmoid(s["colors"][:, :3]))[1]
    s16 = {k: (v.half() if k in ("quats", "scales", "opacities", "colors") else v) for k, v in s.items()}
    s16["colors"] = torch.sigmoid(s["colors"][:, :3]).half()
    img = _run(s16, hip_device, mode="RGB", colors=s16["colors"])[1]
    p = psnr(img.cpu(), ref.cpu())
    print(f"fp16-attribute render vs fp32: {p:.1f} dB")
    assert p > 40.0


def test_800k_fp32_twin_of_config5_against_c_oracle(hip_device, capsys):
    """BASELINE config #5's scene (800 000 Gaussians, seed 1) in fp32 through rasterization(), forward AND backward,
    against the C oracle (the 800 k scene had only been compared with itself).  Same criteria as the
    300 k test: radii, image within 3e-5 of the range except blend flips inside the derived one-step bound, |dPSNR| <=
    1e-4 dB, gradients rtol 1e-3 + 1e-4 of the maximum."""
    import bench as B
    from oracle import gsplat_cpu as Cc
    from oracle import render_torch as R
    from mobgs_amd.rendering import rasterization
    from mobgs_amd.synth import SynthCamera, dynamic_extras, gaussian_cloud
    ns, nd = 533_000, 267_000
    scam = SynthCamera()
    sp, dp = gaussian_cloud(ns, scam, 1), gaussian_cloud(nd, scam, 2)
    dx = dynamic_extras(dp["xyz"], 1)
    ctrl = R.hermite(dx["control_xyz"], torch.tensor(scam.time), dx["current_control_num"]) * 1e-2
    tfp = scam.time - dx["trbf_center"]
    s = {"means": torch.cat([sp["xyz"], ctrl]), "quats": torch.cat([sp["rotation"], dp["rotation"] + tfp * dx["omega"]]),
         "scales": torch.exp(torch.cat([sp["scaling"], dp["scaling"]])),
         "opacities": torch.sigmoid(torch.cat([sp["opacity"], dp["opacity"]])).squeeze(-1),
         "colors": torch.cat([torch.cat([sp["features_dc"], 0 * sp["features_t"]], 1),
                              torch.cat([dp["features_dc"], tfp * dp["features_t"]], 1)]),
         "viewmats": torch.eye(4)[None], "Ks": scam.K[None]}
    names = ["means", "quats", "scales", "opacities", "colors", "viewmats"]
    g = torch.Generator().manual_seed(101)
    v_img = torch.randn(1, H, W, 10, generator=g)
    t = {k: v.to(hip_device).clone().requires_grad_(k in names) for k, v in s.items()}
    img, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], t["viewmats"],
                                 t["Ks"], W, H, packed=False, backgrounds=torch.zeros(1, 9, device=hip_device),
                                 render_mode="RGB+ED")
    (img * v_img.to(hip_device)).sum().backward()
    r = Cc.rasterization_fwd_bwd(*(s[k].numpy() for k in ["means", "quats", "scales", "opacities", "colors",
                                                          "viewmats", "Ks"]), W, H,
                                 backgrounds=np.zeros((1, 9), np.float32), render_mode="RGB+ED",
                                 v_render=v_img.numpy())
    rad = meta["radii"].cpu().numpy().astype(np.int64)
    differ = rad != r["radii"].astype(np.int64)
    ref = torch.from_numpy(r["render"])
    scale = float(ref.abs().max())
    vis_depth = meta["depths"][meta["radii"] > 0]
    nflip, worst = close_image_with_blend_flips(img, ref, r["alphas"], float(s["colors"].abs().max()),
                                                float(vis_depth.max() - vis_depth.min()), 3e-5 * scale,
                                                "image (800k)", flip_frac=2e-4, n_colour_channels=9, alphas_img=a)
    with capsys.disabled():
        print(f"\n[800k fp32] radii differing: {int(differ.sum())} of {rad.size}; image: {nflip} of {ref.numel()} "
              f"elements beyond 3e-5 x range (largest {worst:.2e}); visible {int((rad > 0).sum())}")
    assert differ.sum() == 0, int(differ.sum())
    target = (ref[..., :9] / scale + 0.05 * torch.randn(ref[..., :9].shape, generator=g))
    assert abs(psnr(img[..., :9].cpu() / scale, target) - psnr(ref[..., :9] / scale, target)) <= 1e-4
    for k, ck in [("means", "v_means"), ("quats", "v_quats"), ("scales", "v_scales"), ("opacities", "v_opacities"),
                  ("colors", "v_colors"), ("viewmats", "v_viewmats")]:
        refg = torch.from_numpy(r[ck])
        sc = float(refg.abs().max())
        close(t[k].grad, refg, 1e-3, 1e-4 * sc, f"grad[{k}] (800k)", flip_frac=1e-5, flip_atol=5e-3 * sc)


def test_800k_gaussians_config5_scale(hip_device):
    from mobgs_amd.gaussian_renderer import render
    import bench as B
    scam, cam, stat, dyn, _ = B.build_scene(hip_device, 533_000, 267_000, W, H, seed=1)
    out = render(cam, stat, dyn, None, torch.zeros(9, device=hip_device))
    (out["render"].sum() + out["depth"].sum()).backward()
    assert torch.isfinite(out["render"]).all() and torch.isfinite(stat._xyz.grad).all()
    assert int((out["radii"] > 0).sum()) > 700_000


def test_fullsize_lean_render_against_the_oracle_chain(hip_device, capsys):
    """the BENCHMARK's call -- lean render() forward + backward at 1352x1014, 200 k static + 100 k
    dynamic splats, Morton-sorted rows, a non-trivial pose with its gradient, the library's DEFAULT kernel selection
    for this grid (asserted: project_fwd<PREP> / project_bwd<PREPB>, one wave per tile raster_fwd_blocks<10, ., DECODE>,
    quadrant raster_bwd_kernel<10> with the static-row blend body) -- against a render()-level oracle on the host:
    oracle/render_torch.render (spline, activations, colour features and the Sandwich decoder in plain torch, restating
    /root/reference/gaussian_renderer/__init__.py:59-316 and helper_model.py:19-28) over oracle/gsplat_cpu.c
    (projection, lists, compositing both ways in C).  Decoded image, expected depth, radii, every leaf gradient, the
    decoder-weight gradients, the pose gradient and viewspace_points.grad; blend-flip-aware criteria as everywhere."""
    import bench as B
    from helpers import decoded_flip_bound
    from mobgs_amd import _fast, rendering
    from mobgs_amd.camera import PinholeCamera
    from mobgs_amd.gaussian_model import GaussianParams
    from mobgs_amd.gaussian_renderer import render
    from mobgs_amd.helper_model import Sandwich
    from oracle import gsplat_cpu as Cc
    from oracle import render_torch as R
    dev = hip_device
    ns, nd = 200_000, 100_000
    scam, _, stat, dyn, _ = B.build_scene(dev, ns, nd, W, H, seed=0)
    pose = B.view_pose(1)
    g = torch.Generator().manual_seed(77)
    v_img, v_dep = torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g)

    def params_on(pc, device, dec, dynamic):
        keys = ["xyz", "scaling", "rotation", "opacity", "features_dc", "features_t"]
        base = {k: getattr(pc, "_" + k).detach().cpu() for k in keys if not (dynamic and k == "xyz")}
        if dynamic:
            base["xyz"] = pc.get_xyz.detach().cpu()
            extra = {"omega": pc._omega.detach().cpu(), "trbf_center": pc.get_trbfcenter.detach().cpu(),
                     "control_xyz": pc.get_control_xyz.detach().cpu(),
                     "current_control_num": pc.current_control_num.detach().cpu()}
            return GaussianParams(base, extra, dec, device, requires_grad=True)
        return GaussianParams(base, None, dec, device, requires_grad=True)

    res = {}
    rendering.path_log = []
    try:
        for name, device in (("hip", dev), ("oracle", torch.device("cpu"))):
            dec = Sandwich(9, 3)
            dec.load_state_dict({k: v.detach().cpu() for k, v in dyn.rgbdecoder.state_dict().items()})
            dec = dec.to(device)
            st, dy = params_on(stat, device, dec, False), params_on(dyn, device, dec, True)
            if name == "hip":
                st.rows_coherent, dy.rows_coherent = stat.rows_coherent, dyn.rows_coherent   # (the bench scene's row order)
            cam = PinholeCamera(W, H, scam.K, pose, scam.time, scam.max_time, device=device)
            cam.world_view_transform.requires_grad_(True)
            bg = torch.zeros(9, device=device)
            if name == "hip":
                out = render(cam, st, dy, None, bg)   # frame 0: two-pass lists
                out = render(cam, st, dy, None, bg)   # the steady state the benchmark times: single-pass lists
            else:
                out = R.render(cam, st, dy, bg, rasterization=Cc.torch_rasterization)
            ((out["render"] * v_img.to(device)).sum() + (out["depth"].reshape(1, H, W) * v_dep.to(device)).sum()).backward()
            leaves = {"s_" + k: getattr(st, "_" + k).grad for k in ("xyz", "scaling", "rotation", "opacity", "features_dc",
                                                                    "features_t")}
            leaves.update({"d_" + k: getattr(dy, "_" + k).grad for k in ("scaling", "rotation", "opacity", "features_dc",
                                                                         "features_t", "omega")})
            leaves["d_control_xyz"] = dy.control_xyz.grad
            leaves["w1"], leaves["w2"] = dec.mlp1.weight.grad, dec.mlp2.weight.grad
            leaves["pose"] = cam.world_view_transform.grad
            leaves["viewspace"] = out["viewspace_points"].grad
            res[name] = (out["render"].detach().cpu(), out["depth"].detach().cpu().reshape(1, H, W), out["radii"].cpu(),
                         {k: (None if v is None else v.detach().cpu()) for k, v in leaves.items()},
                         float(R._dyn_state(dy, cam.time, cam.max_time, None)[3].detach().abs().max()) if name == "oracle"
                         else 0.0, dec)
        log = list(rendering.path_log)
    finally:
        rendering.path_log = None
    # the kernels this ran on
    e_f = [e for e in log if e["dir"] == "fwd" and e["D"] == 10][-1]
    e_b = [e for e in log if e["dir"] == "bwd" and e["D"] == 10][-1]
    assert e_f["n_tiles"] == 85 * 64 and e_f["fwd_kernel"] == "blocks" and e_b["bwd_kernel"] == "quadrant", (e_f, e_b)
    assert e_f["heavy_tiles"] <= e_f["n_tiles"] // 8
    if _fast.get() is not None:
        assert e_f["decode"] and [e for e in log if e["dir"] == "prep"][-1]["fused"] and rendering.fused_calls[0] > 0
    if rendering.STATIC_ROWS:
        assert e_b["static_rows"] == ns
    hip, ora = res["hip"], res["oracle"]
    # radii.  (1) The strong statement: the C oracle's projection fed the kernel's OWN activated state (spline position,
    # rotation + t omega, exp, as the prep kernel evaluates them -- bit-identical to what project_fwd<PREP> builds in
    # registers, tests/test_gpu_fused_prep.py) returns the kernel's radii bit for bit, all 300 000.  (2) Against the oracle's
    # own activation (torch exp / sigmoid on the host: an ulp apart from the device's in ~1e-3 of the values) a splat whose
    # 3 sqrt(lambda) sits within that ulp of an integer lands on the other side of ceil(): at most 1e-5 of the splats,
    # by exactly one (observed: 1 of 300 000; VERDICT r5 item 9 asked for this check instead of an allowance alone).
    from mobgs_amd.gaussian_renderer import _prep, _times
    cam_h = PinholeCamera(W, H, scam.K, pose, scam.time, scam.max_time, device=dev)
    with torch.no_grad():
        m, q, sc_, op_, _ = _prep(stat, dyn, _times(cam_h, None, dev))
    rad_chk = Cc.project_fwd(m.cpu().numpy(), q.cpu().numpy(), sc_.cpu().numpy(), pose[None].numpy(),
                             scam.K[None].numpy(), W, H)[0][0]
    assert np.array_equal(rad_chk, hip[2].numpy()), \
        f"radii differ from the oracle fed the kernel's activated state: {int((rad_chk != hip[2].numpy()).sum())}"
    dr = (hip[2].long() - ora[2].long()).abs()
    differ = int((dr != 0).sum())
    assert differ <= max(1, (ns + nd) // 100_000) and int(dr.max()) <= 1, \
        f"radii differing from the oracle chain: {differ} of {ns + nd}, by up to {int(dr.max())}"
    cmax = max(ora[4], float(stat._features_dc.detach().abs().max()))
    fb = decoded_flip_bound(ora[5], cmax)
    nbad = int(((hip[0] - ora[0]).abs() > 3e-5).sum())
    close(hip[0], ora[0], 0, 3e-5, "decoded image", flip_frac=2e-4, flip_atol=fb)
    dref = ora[1]
    vis = float(dref.max() - dref.min())
    close(hip[1], dref, 0, 3e-5 * max(1.0, float(dref.abs().max())), "expected depth", flip_frac=2e-4,
          flip_atol=2.0 * (1.001 / 255.0) * vis / (1.0 / 255.0))
    target = (ora[0] + 0.05 * torch.randn(ora[0].shape, generator=g)).clamp(0, 1)
    assert abs(psnr(hip[0], target) - psnr(ora[0], target)) <= 1e-4
    worst = {}
    for k, ref in ora[3].items():
        got = hip[3][k]
        if ref is None:
            assert got is None or float(got.abs().max()) == 0.0, k
            continue
        if k == "s_features_t":   # 0.0 * f_t: the gradient is 0.0 * v (exact zeros either way)
            assert float(ref.abs().max()) == 0.0 and float(got.abs().max()) == 0.0
            continue
        sc = float(ref.abs().max())
        worst[k] = float((got - ref).abs().max()) / max(sc, 1e-30)
        # the full-size operator test's criterion: rtol 1e-3 + 1e-4 of the maximum; a flipped blend decision moves the
        # gradients of ONE pixel's splats: 1e-5 of the entries up to 5e-3 of the maximum.  Sums over all pixels (decoder
        # weights, pose: 16 / 72 / 18 numbers each fed by 1.4 M pixels of random-sign terms that largely cancel) get 2e-3 of the maximum
        if k in ("w1", "w2", "pose"):   # observed: w1 9e-4 of the maximum in 3 of 72 entries, pose / w2 below 2e-4
            close(got, ref, 2e-3, 2e-3 * sc, f"grad[{k}]")
        else:
            # (2e-5 here against the operator-level test's 1e-5: on top of flipped blend decisions this chain has the
            # splat(s) whose radius differs by one -- other tile rectangle, other pixels reached; observed 3 of 200 000)
            close(got, ref, 1e-3, 1e-4 * sc, f"grad[{k}]", flip_frac=2e-5, flip_atol=5e-3 * sc)
    with capsys.disabled():
        print(f"\n[fullsize render()] radii: bit-equal to the oracle fed the kernel's state; {differ} of {ns + nd} differ "
              f"(by 1) from the oracle's own activation")
        print(f"[fullsize render()] kernels: {e_f['fwd_kernel']} fwd (decode={e_f['decode']}), {e_b['bwd_kernel']} bwd, "
              f"static_rows={e_b.get('static_rows')}, heavy tiles {e_f['heavy_tiles']} of {e_f['n_tiles']}; image elements "
              f"beyond 3e-5: {nbad} of {hip[0].numel()}; largest gradient error / max: "
              + ", ".join(f"{k} {v:.1e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:5]))
