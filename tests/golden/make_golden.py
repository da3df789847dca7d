"""Generate the golden fixtures under tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE in this container.

    python tests/golden/make_golden.py

Needs /root/reference (read-only, never copied) -- see ref_harness.py for how it is made importable on CPU.
The rasterizer under the reference's render()/get_flow() is the CPU oracle oracle/gsplat_torch.py (real gsplat
is absent), so these fixtures PIN the reference's own glue (Hermite spline, rotation/colour build, call order,
decoder, output dict) and only carry the oracle's rasterizer arithmetic along.
Fixtures hold inputs, outputs and autograd gradients only -- data, no reference source.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness as RH  # noqa: E402

from mobgs_amd.camera import PinholeCamera  # noqa: E402
from mobgs_amd.synth import SynthCamera, dynamic_extras, gaussian_cloud  # noqa: E402

STAT_KEYS = ["xyz", "scaling", "rotation", "opacity", "features_dc", "features_t"]
DYN_KEYS = ["omega", "trbf_center", "control_xyz", "current_control_num"]


def small_w2c():
    a, b = 0.04, -0.03
    Ry = torch.tensor([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=torch.float32)
    Rx = torch.tensor([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]], dtype=torch.float32)
    w2c = torch.eye(4)
    w2c[:3, :3] = Ry @ Rx
    w2c[:3, 3] = torch.tensor([0.03, -0.02, 0.05])
    return w2c


def scene_params(ns, nd, cam, seed):
    stat = gaussian_cloud(ns, cam, seed)
    dyn = gaussian_cloud(nd, cam, seed + 1)
    dyn.update(dynamic_extras(dyn["xyz"], seed))
    return stat, dyn


def ref_models(stat, dyn, seed):
    """The reference's own GaussianModel objects filled with the synthetic parameters."""
    gm = RH.ref_import("scene.gaussian_model")
    with RH.CudaToCpu():
        torch.manual_seed(seed)
        spc = gm.GaussianModel(0, RH.Args())
        dpc = gm.GaussianModel(0, RH.Args())

    def fill(pc, p, dynamic):
        pc._xyz = p["xyz"].clone().requires_grad_(True)
        pc._scaling = p["scaling"].clone().requires_grad_(True)
        pc._rotation = p["rotation"].clone().requires_grad_(True)
        pc._opacity = p["opacity"].clone().requires_grad_(True)
        pc._features_dc = p["features_dc"].clone().requires_grad_(True)
        pc._features_t = p["features_t"].clone().requires_grad_(True)
        if dynamic:
            pc._omega = p["omega"].clone().requires_grad_(True)
            pc._trbf_center = p["trbf_center"].clone().requires_grad_(True)
            pc.control_xyz = p["control_xyz"].clone().requires_grad_(True)
            pc.current_control_num = p["current_control_num"].clone()

    fill(spc, stat, False)
    fill(dpc, dyn, True)
    return spc, dpc


def leafs(spc, dpc):
    d = {"s_" + k: getattr(spc, k) for k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc",
                                              "_features_t")}
    d.update({"d_" + k: getattr(dpc, k) for k in ("_scaling", "_rotation", "_opacity", "_features_dc",
                                                  "_features_t", "_omega", "control_xyz")})
    d["w1"] = dpc.rgbdecoder.mlp1.weight
    d["w2"] = dpc.rgbdecoder.mlp2.weight
    return d


def np_(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, **arrays):
    arrays = {k: v for k, v in arrays.items() if v is not None}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays)")


def inputs_dict(stat, dyn, cam, w2c, spc, dpc, bg):
    d = {"in_s_" + k: np_(stat[k]) for k in STAT_KEYS}
    d.update({"in_d_" + k: np_(dyn[k]) for k in STAT_KEYS + DYN_KEYS})
    d.update({"in_w1": np_(dpc.rgbdecoder.mlp1.weight), "in_w2": np_(dpc.rgbdecoder.mlp2.weight),
              "in_w2c": np_(w2c), "in_K": np_(cam.K), "in_bg": np_(bg),
              "in_cam": np.array([cam.image_width, cam.image_height, cam.time, cam.max_time], dtype=np.float64)})
    return d


def gen_render(name, ns, nd, W, H, seed, get_static, get_dynamic, delta, get_flow, use_w2c_arg):
    gr = RH.ref_import("gaussian_renderer")
    scam = SynthCamera().scaled(W, H)
    w2c = small_w2c()
    cam = PinholeCamera(W, H, scam.K, w2c, time=scam.time, max_time=scam.max_time)
    stat, dyn = scene_params(ns, nd, scam, seed)
    spc, dpc = ref_models(stat, dyn, seed)
    bg = torch.tensor([0.1, 0.2, 0.3, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    w2c_leaf = w2c.clone().requires_grad_(True) if use_w2c_arg else None
    delta_t = None if delta is None else torch.tensor(delta)
    with RH.CudaToCpu():
        out = gr.render(cam, spc, dpc, None, bg, get_static=get_static, get_dynamic=get_dynamic, w2c=w2c_leaf,
                        delta_exposure=delta_t, get_flow=get_flow)
    g = torch.Generator().manual_seed(seed + 100)
    v_render = torch.randn(out["render"].shape, generator=g)
    v_depth = torch.randn(out["depth"].shape, generator=g)
    loss = (out["render"] * v_render).sum() + (out["depth"] * v_depth).sum()
    if get_static:
        loss = loss + (out["s_render"] * v_render).sum() * 0.5 + (out["s_alpha"] * v_depth).sum() * 0.25
    if get_dynamic:
        loss = loss + (out["d_render"] * v_render).sum() * 0.5 + (out["d_alpha"] * v_depth).sum() * 0.25 \
            + (out["d_depth"] * v_depth).sum() * 0.125
    loss.backward()
    arrays = inputs_dict(stat, dyn, cam, w2c, spc, dpc, bg)
    arrays["opt"] = np.array([int(get_static), int(get_dynamic), 0 if delta is None else 1,
                              0.0 if delta is None else delta, int(get_flow), int(use_w2c_arg)], dtype=np.float64)
    for k, v in out.items():
        if isinstance(v, torch.Tensor):
            arrays["out_" + k] = np_(v)
    arrays["cot_v_render"] = np_(v_render)
    arrays["cot_v_depth"] = np_(v_depth)
    for k, v in leafs(spc, dpc).items():
        arrays["grad_" + k] = np_(v.grad) if v.grad is not None else None
    if use_w2c_arg:
        arrays["grad_w2c"] = np_(w2c_leaf.grad)
    arrays["grad_viewspace_points"] = np_(out["viewspace_points"].grad)
    save(name, **arrays)


def gen_get_flow(name, ns, nd, W, H, seed, delta):
    gr = RH.ref_import("gaussian_renderer")
    scam = SynthCamera().scaled(W, H)
    w2c = small_w2c()
    cam = PinholeCamera(W, H, scam.K, w2c, time=scam.time, max_time=scam.max_time)
    stat, dyn = scene_params(ns, nd, scam, seed)
    spc, dpc = ref_models(stat, dyn, seed)
    bg = torch.zeros(9)
    with RH.CudaToCpu(), torch.no_grad():
        e2m, m2e, img, alpha = gr.get_flow(cam, spc, dpc, None, bg, delta_exposure=torch.tensor(delta))
        # get_flow_static: source / target / splat cameras differ by a small pose change
        w2c_b = w2c.clone()
        w2c_b[:3, 3] += torch.tensor([0.02, 0.01, -0.01])
        cam_b = PinholeCamera(W, H, scam.K, w2c_b, time=scam.time, max_time=scam.max_time)
        flow_2d, flow_img = gr.get_flow_static(cam, cam_b, cam, spc, dpc, None, bg)
    arrays = inputs_dict(stat, dyn, cam, w2c, spc, dpc, bg)
    arrays.update({"opt": np.array([delta]), "in_w2c_b": np_(w2c_b), "out_exp2mid": np_(e2m), "out_mid2exp": np_(m2e),
                   "out_latent_img": np_(img), "out_latent_alpha": np_(alpha), "out_static_flow_2d": np_(flow_2d),
                   "out_static_flow_img": np_(flow_img)})
    save(name, **arrays)


def gen_get_flow_grad(name, ns, nd, W, H, seed, delta):
    """get_flow() / get_flow_static() WITH gradients (train.py:570-579 back-propagates the flow loss through them):
    fixed cotangents on all four outputs -> gradients of every leaf that receives one."""
    gr = RH.ref_import("gaussian_renderer")
    scam = SynthCamera().scaled(W, H)
    w2c = small_w2c()
    cam = PinholeCamera(W, H, scam.K, w2c, time=scam.time, max_time=scam.max_time)
    stat, dyn = scene_params(ns, nd, scam, seed)
    spc, dpc = ref_models(stat, dyn, seed)
    bg = torch.zeros(9)
    with RH.CudaToCpu():
        e2m, m2e, img, alpha = gr.get_flow(cam, spc, dpc, None, bg, delta_exposure=torch.tensor(delta))
    g = torch.Generator().manual_seed(seed + 100)
    cots = [torch.randn(t.shape, generator=g) for t in (e2m, m2e, img, alpha)]
    torch.autograd.backward([e2m, m2e, img, alpha], cots)
    arrays = inputs_dict(stat, dyn, cam, w2c, spc, dpc, bg)
    arrays.update({"opt": np.array([delta]), "out_exp2mid": np_(e2m), "out_mid2exp": np_(m2e),
                   "out_latent_img": np_(img), "out_latent_alpha": np_(alpha)})
    for n_, c in zip(("exp2mid", "mid2exp", "latent_img", "latent_alpha"), cots):
        arrays["cot_" + n_] = np_(c)
    for k, v in leafs(spc, dpc).items():
        arrays["grad_" + k] = np_(v.grad) if v.grad is not None else None
        v.grad = None
    # get_flow_static
    w2c_b = w2c.clone()
    w2c_b[:3, 3] += torch.tensor([0.02, 0.01, -0.01])
    cam_b = PinholeCamera(W, H, scam.K, w2c_b, time=scam.time, max_time=scam.max_time)
    with RH.CudaToCpu():
        flow_2d, flow_img = gr.get_flow_static(cam, cam_b, cam, spc, dpc, None, bg)
    c2 = [torch.randn(t.shape, generator=g) for t in (flow_2d, flow_img)]
    torch.autograd.backward([flow_2d, flow_img], c2)
    arrays.update({"in_w2c_b": np_(w2c_b), "out_static_flow_2d": np_(flow_2d), "out_static_flow_img": np_(flow_img),
                   "cot_static_flow_2d": np_(c2[0]), "cot_static_flow_img": np_(c2[1])})
    for k, v in leafs(spc, dpc).items():
        arrays["sgrad_" + k] = np_(v.grad) if v.grad is not None else None
    save(name, **arrays)


def gen_hermite(name):
    gr = RH.ref_import("gaussian_renderer")
    g = torch.Generator().manual_seed(11)
    n = 400
    ctrl = torch.randn(n, 12, 3, generator=g).requires_grad_(True)
    ncp = torch.randint(4, 13, (n, 1), generator=g, dtype=torch.int64)
    ts = [0.0, 1.0, 0.5, 11.0 / 23.0, 0.999999, 1e-7, 1.0 / 3.0, 0.25]
    outs, grads = [], []
    for t in ts:
        with RH.CudaToCpu():
            o = gr.interpolate_cubic_hermite(ctrl.permute(0, 2, 1), torch.tensor(t)[None, None].expand(n, 3, 1), N=ncp)
        v = torch.randn(o.shape, generator=torch.Generator().manual_seed(3))
        gr_, = torch.autograd.grad((o * v).sum(), ctrl)
        outs.append(np_(o))
        grads.append(np_(gr_))
    save(name, control=np_(ctrl), ncp=np_(ncp), ts=np.array(ts, dtype=np.float64), out=np.stack(outs),
         grad=np.stack(grads), cot=np_(torch.randn(n, 3, generator=torch.Generator().manual_seed(3))))


def gen_sandwich(name):
    hm = RH.ref_import("helper_model")
    torch.manual_seed(5)
    dec = hm.Sandwich(9, 3)
    g = torch.Generator().manual_seed(6)
    feat = torch.randn(1, 9, 20, 24, generator=g).requires_grad_(True)
    rays = torch.randn(1, 6, 20, 24, generator=g).requires_grad_(True)
    out = dec(feat, rays)
    v = torch.randn(out.shape, generator=g)
    (out * v).sum().backward()
    save(name, feat=np_(feat), rays=np_(rays), w1=np_(dec.mlp1.weight), w2=np_(dec.mlp2.weight), out=np_(out),
         cot=np_(v), grad_feat=np_(feat.grad), grad_rays=np_(rays.grad), grad_w1=np_(dec.mlp1.weight.grad),
         grad_w2=np_(dec.mlp2.weight.grad))


def deform_weights(net):
    """Flat weight dict of a reference deform_network (state_dict key mapping used by mobgs_amd.deformation)."""
    d = net.deformation_net
    W = {"w0": d.feature_out[0].weight, "b0": d.feature_out[0].bias}
    for name, seq in (("pos", d.pos_deform), ("scl", d.scales_deform), ("rot", d.rotations_deform)):
        W[name + "_w1"], W[name + "_b1"] = seq[1].weight, seq[1].bias
        W[name + "_w2"], W[name + "_b2"] = seq[3].weight, seq[3].bias
    return W


def gen_deform(name, n=700, seed=4):
    dm = RH.ref_import("scene.deformation")
    args = RH.Args()
    args.kplanes_config = dict(args.kplanes_config, resolution=[8, 8, 8, 4])  # small planes keep the fixture small
    with RH.CudaToCpu():
        torch.manual_seed(seed)
        net = dm.deform_network(args)
        # biases keep PyTorch's default init upstream; randomise the planes so the product is not trivial
        g = torch.Generator().manual_seed(seed)
        for level in net.deformation_net.grid.grids:
            for pl in level:
                pl.data = 0.5 + 0.5 * torch.rand(pl.shape, generator=g)
        xyz_max, xyz_min = [1.2, 1.0, 1.5], [-1.1, -0.9, -0.2]
        net.deformation_net.set_aabb(xyz_max, xyz_min)
    g = torch.Generator().manual_seed(seed + 1)
    lo, hi = torch.tensor(xyz_min), torch.tensor(xyz_max)
    pts = (lo + (hi - lo) * (1.2 * torch.rand(n, 3, generator=g) - 0.1)).requires_grad_(True)  # some outside the box
    scales = (0.1 * torch.randn(n, 3, generator=g)).requires_grad_(True)
    rots = torch.randn(n, 4, generator=g).requires_grad_(True)
    times = torch.rand(n, 1, generator=g)
    with RH.CudaToCpu():
        o_pts, o_scl, o_rot = net(pts, scales, rots, times)
    v = [torch.randn(o.shape, generator=g) for o in (o_pts, o_scl, o_rot)]
    ((o_pts * v[0]).sum() + (o_scl * v[1]).sum() + (o_rot * v[2]).sum()).backward()
    arrays = {"in_pts": np_(pts), "in_scales": np_(scales), "in_rots": np_(rots), "in_times": np_(times),
              "in_aabb": np_(net.deformation_net.grid.aabb), "out_pts": np_(o_pts), "out_scales": np_(o_scl),
              "out_rots": np_(o_rot), "cot_pts": np_(v[0]), "cot_scales": np_(v[1]), "cot_rots": np_(v[2]),
              "grad_pts": np_(pts.grad), "grad_scales": np_(scales.grad), "grad_rots": np_(rots.grad)}
    for k, w in deform_weights(net).items():
        arrays["w_" + k] = np_(w)
        arrays["gw_" + k] = np_(w.grad)
    for li, level in enumerate(net.deformation_net.grid.grids):
        for pi, pl in enumerate(level):
            arrays[f"plane_{li}_{pi}"] = np_(pl)
            arrays[f"gplane_{li}_{pi}"] = np_(pl.grad)
    arrays["n_params"] = np.array([sum(p.numel() for p in net.parameters())])
    save(name, **arrays)


def mid_planes(planes, seed):
    """Plane values of the mid-size deformation fixture: drawn from a seeded generator in plane order so that the
    tests regenerate them instead of loading 8 MB (shared with tests/test_gpu_config3.py)."""
    g = torch.Generator().manual_seed(seed)
    return [0.5 + 0.5 * torch.rand(tuple(pl.shape), generator=g) for pl in planes]


def gen_deform_mid(name, n=4000, seed=14, base=32):
    """deform_network of the reference at planes [32,32,32,12] x multires [1,2,4] (half the seesaw resolution per
    axis; the toy fixture `deform` has 8): outputs, input gradients, all MLP weight gradients, the 9 time-plane
    gradients in full, and for the 9 spatial planes {sum, sum |.|, 2048 seeded samples} of the gradient."""
    dm = RH.ref_import("scene.deformation")
    args = RH.Args()
    args.kplanes_config = dict(args.kplanes_config, resolution=[base, base, base, 12])
    with RH.CudaToCpu():
        torch.manual_seed(seed)
        net = dm.deform_network(args)
        planes = [pl for level in net.deformation_net.grid.grids for pl in level]
        for pl, v in zip(planes, mid_planes(planes, seed)):
            pl.data = v
        xyz_max, xyz_min = [1.3, 0.9, 1.6], [-1.0, -1.1, -0.3]
        net.deformation_net.set_aabb(xyz_max, xyz_min)
        with torch.no_grad():  # Xavier weights give sub-millimetre deformations: scale the heads up
            for p_ in net.deformation_net.get_mlp_parameters():
                p_.mul_(2.0)
    g = torch.Generator().manual_seed(seed + 1)
    lo, hi = torch.tensor(xyz_min), torch.tensor(xyz_max)
    pts = (lo + (hi - lo) * (1.1 * torch.rand(n, 3, generator=g) - 0.05)).requires_grad_(True)
    scales = (0.1 * torch.randn(n, 3, generator=g)).requires_grad_(True)
    rots = torch.randn(n, 4, generator=g).requires_grad_(True)
    times = torch.full((n, 1), 11.0 / 23.0)  # one time stamp per call, as a training view has
    times[: n // 8] = torch.rand(n // 8, 1, generator=g)  # ... and a block of per-point times (API allows it)
    with RH.CudaToCpu():
        o_pts, o_scl, o_rot = net(pts, scales, rots, times)
    v = [torch.randn(o.shape, generator=g) for o in (o_pts, o_scl, o_rot)]
    ((o_pts * v[0]).sum() + (o_scl * v[1]).sum() + (o_rot * v[2]).sum()).backward()
    arrays = {"in_pts": np_(pts), "in_scales": np_(scales), "in_rots": np_(rots), "in_times": np_(times),
              "in_aabb": np_(net.deformation_net.grid.aabb), "out_pts": np_(o_pts), "out_scales": np_(o_scl),
              "out_rots": np_(o_rot), "cot_pts": np_(v[0]), "cot_scales": np_(v[1]), "cot_rots": np_(v[2]),
              "grad_pts": np_(pts.grad), "grad_scales": np_(scales.grad), "grad_rots": np_(rots.grad),
              "meta": np.array([seed, base, n])}
    for k, w in deform_weights(net).items():
        arrays["w_" + k] = np_(w)
        arrays["gw_" + k] = np_(w.grad)
    gs = torch.Generator().manual_seed(seed + 2)
    for li, level in enumerate(net.deformation_net.grid.grids):
        for pi, pl in enumerate(level):
            gr = pl.grad.reshape(-1)
            if pi in (2, 4, 5):  # planes with the time axis: small, stored in full
                arrays[f"gplane_{li}_{pi}"] = np_(pl.grad)
            else:
                idx = torch.randint(0, gr.numel(), (2048,), generator=gs)
                arrays[f"gplane_idx_{li}_{pi}"] = idx.numpy()
                arrays[f"gplane_val_{li}_{pi}"] = np_(gr[idx])
                arrays[f"gplane_sum_{li}_{pi}"] = np.array([float(gr.double().sum()), float(gr.double().abs().sum())])
    save(name, **arrays)


def gen_blce(name, num_views=3, seed=8):
    bl = RH.ref_import("scene.blce")
    with RH.CudaToCpu():
        torch.manual_seed(seed)
        model = bl.BLCE(num_views=num_views, view_dim=32, num_warp=9, method="euler", adjoint=False)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():  # the decoders start at 1e-5 gain (near-identity warps): make the fixture non-trivial
            for i in range(num_views):
                for dec, sc in ((model.rot_decoder[i], 0.3), (model.trans_decoder[i], 0.05), (model.theta_decoder[i], 0.1)):
                    dec.weight.copy_(sc * torch.randn(dec.weight.shape, generator=g))
                    dec.bias.copy_(0.1 * sc * torch.randn(dec.bias.shape, generator=g))
                model.wv_derivative[i].time_embedder.data = 0.5 * torch.randn(9, 8, generator=g)
            model.view_embedder.data = torch.randn(num_views, 32, generator=g)
            model.exposure_time_expo.data = torch.tensor([0.4, 0.25, 0.6])[:num_views]
        c2w = torch.inverse(small_w2c())
        img = torch.rand(3, 48, 64, generator=g)
        blur = bl.compute_frequency_blur_feature(img)
        idx = 1
        Rt_new, expo = model(c2w, blur, idx)
    v = torch.randn(Rt_new.shape, generator=g)
    (Rt_new * v).sum().backward()
    arrays = {"in_c2w": np_(c2w), "in_image": np_(img), "in_idx": np.array([idx]), "out_blur": np_(blur),
              "out_Rt_new": np_(Rt_new), "out_exposure": np_(expo), "cot": np_(v),
              "n_params": np.array([sum(p.numel() for p in model.parameters())])}
    for k, p_ in model.state_dict().items():
        arrays["sd_" + k] = np_(p_)
    for k, p_ in model.named_parameters():
        if p_.grad is not None and (f".{idx}." in k or k == "view_embedder"):
            arrays["grad_" + k] = np_(p_.grad)
    save(name, **arrays)


def gen_blurry_view(name, ns=700, nd=400, W=64, H=48, seed=31, num_views=2):
    """One blurry training view as train.py:441-541 forms it, with the reference's own pieces: scene.blce.BLCE.forward
    -> 9 latent c2w poses + exposure offsets; blceKernel.get_warped_cams' pose algebra (:150-157: R = c2w rotation,
    T = translation of the inverse); gaussian_renderer.render() for the mid frame (train mode, the view's own camera)
    and the 8 latent frames (warped camera, delta_exposure = exposure_time[k]); mean + 1e-10.  Back-propagated:
    sum(pred * v) + sum(depth_mid * vd) + sum(d_alpha_mid * vd) -> gradients of every Gaussian leaf, the decoder and
    the BLCE parameters of that view."""
    gr = RH.ref_import("gaussian_renderer")
    bl = RH.ref_import("scene.blce")
    scam = SynthCamera().scaled(W, H)
    w2c = small_w2c()
    cam = PinholeCamera(W, H, scam.K, w2c, time=scam.time, max_time=scam.max_time)
    stat, dyn = scene_params(ns, nd, scam, seed)
    spc, dpc = ref_models(stat, dyn, seed)
    bg = torch.tensor([0.1, 0.2, 0.3, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    idx = 1
    with RH.CudaToCpu():
        torch.manual_seed(seed)
        model = bl.BLCE(num_views=num_views, view_dim=32, num_warp=9, method="euler", adjoint=False)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():  # decoders start at 1e-5 gain: give the latent poses a visible spread
            for i in range(num_views):
                for dec, sc in ((model.rot_decoder[i], 0.3), (model.trans_decoder[i], 0.02), (model.theta_decoder[i], 0.03)):
                    dec.weight.copy_(sc * torch.randn(dec.weight.shape, generator=g))
                    dec.bias.copy_(0.1 * sc * torch.randn(dec.bias.shape, generator=g))
                model.wv_derivative[i].time_embedder.data = 0.5 * torch.randn(9, 8, generator=g)
            model.view_embedder.data = torch.randn(num_views, 32, generator=g)
            model.exposure_time_expo.data = torch.tensor([0.4, 0.3, 0.6])[:num_views]
        img = torch.rand(3, H, W, generator=g)
        blur = bl.compute_frequency_blur_feature(img)
        c2w = torch.inverse(w2c)
        warped_c2w, expo = model(c2w, blur, idx)
        warped_w2c_full = torch.inverse(warped_c2w)
        renders = []
        mid = gr.render(cam, spc, dpc, None, bg, get_static=True, get_dynamic=True)
        for k in range(9):
            if k == 4:
                renders.append(mid["render"])
                continue
            # get_cam (:160-163): Camera(R = c2w rotation, T = w2c translation) -> world_view_transform = [R^T | T]
            R, T = warped_c2w[k, :3, :3], warped_w2c_full[k, :3, 3]
            w2c_k = torch.eye(4)
            w2c_k = torch.cat([torch.cat([R.transpose(0, 1), T[:, None]], dim=1), torch.tensor([[0.0, 0, 0, 1]])], dim=0)
            cam_k = PinholeCamera(W, H, scam.K, w2c_k.detach(), time=scam.time, max_time=scam.max_time)
            cam_k.world_view_transform = w2c_k.transpose(0, 1)  # differentiable pose (Camera keeps tensors, :109-146)
            c2w_k = torch.inverse(w2c_k)
            cam_k._ray = PinholeCamera.build_cam_ray_c2w(W, H, scam.K, c2w_k)
            pkg = gr.render(cam_k, spc, dpc, None, bg, get_static=True, get_dynamic=True, delta_exposure=expo[k])
            renders.append(pkg["render"])
        pred = torch.mean(torch.stack(renders, dim=0), dim=0) + 1e-10
    gq = torch.Generator().manual_seed(seed + 100)
    v_pred = torch.randn(pred.shape, generator=gq)
    v_depth = torch.randn(mid["depth"].shape, generator=gq)
    ((pred * v_pred).sum() + (mid["depth"] * v_depth).sum() + (mid["d_alpha"] * v_depth).sum()).backward()
    arrays = inputs_dict(stat, dyn, cam, w2c, spc, dpc, bg)
    arrays.update({"in_image": np_(img), "in_idx": np.array([idx, num_views]), "out_pred": np_(pred),
                   "out_mid_render": np_(mid["render"]), "out_mid_depth": np_(mid["depth"]),
                   "out_mid_d_alpha": np_(mid["d_alpha"]), "out_warped_c2w": np_(warped_c2w), "out_exposure": np_(expo),
                   "cot_v_pred": np_(v_pred), "cot_v_depth": np_(v_depth),
                   "grad_viewspace_points": np_(mid["viewspace_points"].grad), "out_radii": np_(mid["radii"])})
    for k, v in leafs(spc, dpc).items():
        arrays["grad_" + k] = np_(v.grad) if v.grad is not None else None
    for k, v in model.state_dict().items():
        arrays["sd_" + k] = np_(v)
    for k, p_ in model.named_parameters():
        if p_.grad is not None and float(p_.grad.abs().max()) > 0:
            arrays["bgrad_" + k] = np_(p_.grad)
    save(name, **arrays)


def gen_losses(name):
    lu = RH.ref_import("utils.loss_utils")
    iu = RH.ref_import("utils.image_utils")
    g = torch.Generator().manual_seed(12)
    H, W = 45, 71  # not multiples of the 16-pixel tile: border handling
    gt = torch.rand(2, 3, H, W, generator=g)
    img = (gt + 0.15 * torch.randn(2, 3, H, W, generator=g)).clamp(0, 1).requires_grad_(True)
    with RH.CudaToCpu():
        l1 = lu.l1_loss(img, gt)
        s = lu.ssim(img, gt)
        s_per = lu.ssim(img, gt, size_average=False)
        loss = l1 + 0.2 * (1.0 - s)
        p = iu.psnr(img.detach(), gt)
    loss.backward()
    save(name, img=np_(img), gt=np_(gt), l1=np_(l1), ssim=np_(s), ssim_per_image=np_(s_per), loss=np_(loss),
         grad_img=np_(img.grad), psnr=np_(p))


# per-splat optimiser groups of the reference (scene/gaussian_model.py:598-617) and the attribute each one backs
DENSIFY_GROUPS = [("xyz", "_xyz"), ("control_xyz", "control_xyz"), ("current_control_num", "current_control_num"),
                  ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("f_t", "_features_t"),
                  ("opacity", "_opacity"), ("scaling", "_scaling"), ("rotation", "_rotation"), ("omega", "_omega"),
                  ("zeta", "_zeta"), ("trbf_center", "_trbf_center"), ("trbf_scale", "_trbf_scale"),
                  ("motion", "_motion")]
DENSIFY_AUX = ["xyz_gradient_accum", "denom", "max_radii2D", "_deformation_table", "_deformation_accum"]


class OptArgs:
    """OptimizationParams defaults (arguments/__init__.py:117-185) read by GaussianModel.training_setup."""
    percent_dense = 0.01
    position_lr_init = 0.00016
    position_lr_final = 0.0000016
    position_lr_max_steps = 20_000
    deformation_lr_init = 0.00016
    deformation_lr_final = 0.000016
    grid_lr_init = 0.0016
    grid_lr_final = 0.00016
    feature_lr = 0.0025
    featuret_lr = 0.001
    opacity_lr = 0.05
    scaling_lr = 0.005
    rotation_lr = 0.001
    omega_lr = 0.0001
    zeta_lr = 0.0001
    trbfc_lr = 0.0001
    trbfs_lr = 0.03
    movelr = 3.5
    rgb_lr = 0.0001
    pose_lr_init = 0.0005
    pose_lr_final = 0.00005


def gen_densify(name, n=600, seed=21):
    """GaussianModel's densification / optimiser surgery (scene/gaussian_model.py:897-904, :1029-1244, :1352-1356,
    :1417-1434, :1480-1506) on a dynamic model with a populated Adam state.  Recorded: every per-splat parameter,
    its exp_avg / exp_avg_sq, and the per-splat statistics after each operation."""
    gm = RH.ref_import("scene.gaussian_model")
    cam = SynthCamera(96, 64)
    g = torch.Generator().manual_seed(seed)
    p = gaussian_cloud(n, cam, seed)
    p.update(dynamic_extras(p["xyz"], seed))
    with RH.CudaToCpu():
        torch.manual_seed(seed)
        pc = gm.GaussianModel(0, RH.Args())
        P = torch.nn.Parameter
        pc._xyz = P(p["xyz"].clone())
        pc._scaling = P(p["scaling"].clone() + 1.2 * torch.randn(n, 3, generator=g))  # both sides of percent_dense
        pc._rotation = P(p["rotation"].clone())
        pc._opacity = P(p["opacity"].reshape(n, 1).clone())
        pc._features_dc = P(p["features_dc"].clone())
        pc._features_rest = P(torch.zeros(n, 0, 3))
        pc._features_t = P(p["features_t"].clone())
        pc._omega = P(p["omega"].clone())
        pc._zeta = P(0.1 * torch.randn(n, 1, generator=g))
        pc._trbf_center = P(p["trbf_center"].clone())
        pc._trbf_scale = P(0.1 * torch.randn(n, 1, generator=g))
        pc._motion = P(0.1 * torch.randn(n, 9, generator=g))
        pc.control_xyz = P(p["control_xyz"].clone())
        pc.current_control_num = p["current_control_num"].clone().reshape(n, 1)
        pc._deformation_table = torch.rand(n, generator=g) > 0.3
        pc.max_radii2D = torch.zeros(n)
        pc.spatial_lr_scale = 1.0
        pc.training_setup(OptArgs())
        # the reference hands current_control_num to Adam as a plain tensor; populate the Adam state with one step
        for gname, attr in DENSIFY_GROUPS:
            t = getattr(pc, attr)
            if t.requires_grad and t.numel() > 0:
                t.grad = torch.randn(t.shape, generator=g) * 0.01
        pc.optimizer.step()

        def state(tag):
            out = {}
            groups = {gr["name"]: gr for gr in pc.optimizer.param_groups}
            for gname, attr in DENSIFY_GROUPS:
                t = getattr(pc, attr)
                assert groups[gname]["params"][0] is t, gname  # the attribute IS the optimiser's parameter
                out[f"{tag}.{gname}"] = np_(t).copy()
                st = pc.optimizer.state.get(t, None)
                if st is not None and "exp_avg" in st:
                    out[f"{tag}.{gname}.exp_avg"] = np_(st["exp_avg"]).copy()
                    out[f"{tag}.{gname}.exp_avg_sq"] = np_(st["exp_avg_sq"]).copy()
            for a in DENSIFY_AUX:
                out[f"{tag}.{a}"] = np_(getattr(pc, a)).copy()
            return out

        rec = {}
        rec.update(state("s0"))
        # the reference's own save_ply (scene/gaussian_model.py:761-804) with plyfile replaced by a recorder:
        # property names in file order and the row matrix it hands to PlyElement.describe
        captured = {}

        class _Elem:
            @staticmethod
            def describe(elements, name):
                captured["names"], captured["element"] = list(elements.dtype.names), name
                captured["rows"] = np.stack([elements[k] for k in elements.dtype.names], 1).astype(np.float32)
                return None

        class _Data:
            def __init__(self, els):
                pass

            def write(self, path):
                captured["path"] = path

        old = gm.PlyElement, gm.PlyData, gm.torch.save, gm.mkdir_p
        gm.PlyElement, gm.PlyData, gm.torch.save, gm.mkdir_p = _Elem, _Data, (lambda *a, **k: None), (lambda *a: None)
        try:
            pc.save_ply("/tmp/_mobgs_golden/point_cloud.ply")
        finally:
            gm.PlyElement, gm.PlyData, gm.torch.save, gm.mkdir_p = old
        rec["ply.names"] = np.array(captured["names"])
        rec["ply.rows"] = captured["rows"]
        rec["ply.element"] = np.array(captured["element"])
        # two iterations of per-step statistics (helper_train.py:263-264)
        stats_in = []
        for it in range(2):
            vsp = torch.randn(n, 3, generator=g) * 3e-4
            vis = torch.rand(n, generator=g) > 0.4
            radii = torch.randint(0, 40, (n,), generator=g).to(torch.float32)
            pc.max_radii2D[vis] = torch.max(pc.max_radii2D[vis], radii[vis])
            pc.add_densification_stats(vsp, vis)
            stats_in.append((vsp, vis, radii))
        rec.update(state("s1"))
        for it, (vsp, vis, radii) in enumerate(stats_in):
            rec[f"stats{it}.viewspace_grad"], rec[f"stats{it}.visible"], rec[f"stats{it}.radii"] = \
                np_(vsp), np_(vis), np_(radii)
        # densify_pruneclone = clone + splitv2 (scene/gaussian_model.py:1417-1434), grads as formed there
        grads = pc.xyz_gradient_accum / pc.denom
        grads[grads.isnan()] = 0.0
        max_grad, extent = 2.0e-4, 4.0
        rec["grads"], rec["max_grad"], rec["extent"] = np_(grads), np.float32(max_grad), np.float32(extent)
        pc.densify_and_clone(grads, max_grad, extent)
        rec.update(state("s2"))
        drawn = {}
        real_normal = torch.normal

        def recording_normal(mean, std, **kw):
            drawn["samples"] = real_normal(mean=mean, std=std, generator=g)
            return drawn["samples"]

        torch.normal = recording_normal
        try:
            pc.densify_and_splitv2(grads, max_grad, extent, 2)
        finally:
            torch.normal = real_normal
        rec["split.samples"] = np_(drawn["samples"])
        rec.update(state("s3"))
        prune_mask = (pc.get_opacity < 0.3).squeeze()
        rec["prune.mask"] = np_(prune_mask)
        pc.prune_points(prune_mask)
        rec.update(state("s4"))
        pc.reset_opacity()
        rec.update(state("s5"))
    save(name, **rec)


def gen_normals(name):
    """main_utils.get_normals (main_utils.py:95-141) on a small depth map, with its gradient w.r.t. the depth."""
    import types as _t
    for mod in ("matplotlib", "PIL"):
        if mod not in sys.modules:
            sys.modules[mod] = _t.ModuleType(mod)
    sys.modules["matplotlib"].cm = getattr(sys.modules["matplotlib"], "cm", None)
    sys.modules["PIL"].Image = getattr(sys.modules["PIL"], "Image", None)
    mu = RH.ref_import("main_utils")
    H, W = 37, 53

    class Meta:  # what get_normals reads from a dycheck camera (dycheck_geometry/camera.py:600-613 for get_pixels)
        principal_point_x, principal_point_y = 25.3, 19.1
        scale_factor_x, scale_factor_y = 61.0, 58.5
        skew = 0.02
        use_center = True
        image_size_x, image_size_y = W, H

        def get_pixels(self, use_center=None, normalize=False):
            xx, yy = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
            return np.stack([xx, yy], axis=-1) + 0.5

    g = torch.Generator().manual_seed(17)
    z = (2.0 + torch.rand(1, H, W, generator=g) + 0.3 * torch.randn(1, H, W, generator=g).abs()).requires_grad_(True)
    z.data[0, 5:9, 7:12] = 3.0  # a flat patch: cross product from exactly equal depths
    with RH.CudaToCpu():
        n = mu.get_normals(z + 1e-6, Meta())
    w = torch.randn(n.shape, generator=g)
    (n * w).sum().backward()
    save(name, z=np_(z), normals=np_(n), cotangent=np_(w), grad_z=np_(z.grad),
         intrinsics=np.array([Meta.scale_factor_x, Meta.scale_factor_y, Meta.principal_point_x,
                              Meta.principal_point_y, Meta.skew], dtype=np.float32))


def main():
    RH.install()
    gen_hermite("hermite")
    gen_sandwich("sandwich")
    gen_render("render_lean", 1200, 600, 96, 64, 0, False, False, None, False, True)
    gen_render("render_train_delta_flow", 900, 500, 80, 48, 1, True, True, 0.3, True, False)
    gen_render("render_train", 700, 400, 64, 48, 2, True, True, None, False, False)
    gen_get_flow("get_flow", 900, 500, 80, 48, 3, -0.4)
    gen_get_flow_grad("get_flow_grad", 800, 450, 72, 48, 6, 0.3)
    gen_deform("deform")
    gen_deform_mid("deform_mid")
    gen_blce("blce")
    gen_blurry_view("blurry_view")
    gen_losses("losses")
    gen_densify("densify")
    gen_normals("normals")


if __name__ == "__main__":
    main()
