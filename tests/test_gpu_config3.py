"""BASELINE config #3 on the GPU: `deform_network` (HexPlane + MLP heads) at the seesaw plane resolution, alone and
feeding the rasterizer.

  * mid-size planes [32,32,32,12] x [1,2,4]: against tests/golden/deform_mid.npz, produced by the reference's own
    deform_network (outputs, input gradients, every MLP weight gradient, time-plane gradients in full, spatial-plane
    gradients by samples + sums);
  * full seesaw planes [64,64,64,12] x [1,2,4] (arguments/stereo/seesaw.py), 100 000 points: against
    oracle/deform_torch.py run live on the host (the restatement is pinned by the two fixtures on CPU);
  * the composed workload: 200 000 static + 100 000 deformed dynamic splats, 1352x1014, rasterization() forward +
    backward down to the deformation network's weights and planes, against deform_torch -> oracle/gsplat_cpu.c ->
    torch autograd through deform_torch.
"""
import numpy as np
import pytest
import torch

from helpers import (check_deform_fixture_grads, close, close_image_with_blend_flips, close_point_rows, load,
                     mid_planes, psnr)

pytestmark = pytest.mark.gpu


def _weights_of(net):
    d = net.deformation_net
    W = {"w0": d.feature_out[0].weight, "b0": d.feature_out[0].bias}
    for name, seq in (("pos", d.pos_deform), ("scl", d.scales_deform), ("rot", d.rotations_deform)):
        W[name + "_w1"], W[name + "_b1"] = seq[1].weight, seq[1].bias
        W[name + "_w2"], W[name + "_b2"] = seq[3].weight, seq[3].bias
    return W


def _make_net(dev, base, seed):
    from mobgs_amd.deformation import SeesawArgs, deform_network

    class A(SeesawArgs):
        kplanes_config = dict(SeesawArgs.kplanes_config, resolution=[base, base, base, 12])

    torch.manual_seed(seed)
    return deform_network(A()).to(dev)


def test_deform_mid_size_planes_match_reference_fixture(hip_device):
    fx = load("deform_mid")
    seed, base, _ = (int(v) for v in fx["meta"])
    net = _make_net(hip_device, base, 0)
    T = lambda k: torch.from_numpy(fx[k]).to(hip_device)  # noqa: E731
    planes = net.deformation_net.grid.planes()
    with torch.no_grad():
        for k, w in _weights_of(net).items():
            w.copy_(T("w_" + k))
        for pl, v in zip(planes, mid_planes([p.shape for p in planes], seed)):
            pl.copy_(v)
        net.deformation_net.grid.aabb.copy_(T("in_aabb"))
    assert all(p.is_contiguous(memory_format=torch.channels_last) for p in planes), "planes must stay channels-last"
    pts, scales, rots = (T(k).requires_grad_(True) for k in ("in_pts", "in_scales", "in_rots"))
    o_pts, o_scl, o_rot = net(pts, scales, rots, T("in_times"))
    close(o_pts, fx["out_pts"], 2e-5, 2e-5, "pts")
    close(o_scl, fx["out_scales"], 2e-5, 2e-5, "scales")
    close(o_rot, fx["out_rots"], 2e-5, 2e-5, "rotations")
    ((o_pts * T("cot_pts")).sum() + (o_scl * T("cot_scales")).sum() + (o_rot * T("cot_rots")).sum()).backward()
    # a hidden unit whose pre-activation is zero to within fp32 rounding takes the other branch of its ReLU in one
    # of the two implementations (here: 1 point of 4000): its three position gradients move by a few per cent
    for name, t in (("pts", pts), ("scales", scales), ("rots", rots)):
        ref = fx["grad_" + name]
        sc = float(np.abs(ref).max())
        close_point_rows(t.grad, ref, 1e-3, 1e-4 * sc, f"grad {name}", flip_rows=1e-3, share=0.25)  # observed: 1 row of 4000
    # ... and so does the row / column of the weight gradients that belongs to that unit (one point's contribution)
    for k, w in _weights_of(net).items():
        ref = fx["gw_" + k]
        sc = float(np.abs(ref).max())
        close(w.grad, ref, 1e-3, 1e-4 * sc + 1e-6, f"grad {k}", flip_frac=0.02, flip_atol=5e-3 * sc)
    grids = net.deformation_net.grid.grids
    check_deform_fixture_grads(fx, lambda li, pi: grids[li][pi].grad)


def _oracle_copy(net):
    """CPU leaves holding the same values as the HIP network (planes in the oracle's [level][plane] nesting)."""
    W = {k: w.detach().cpu().clone().requires_grad_(True) for k, w in _weights_of(net).items()}
    planes = [[pl.detach().cpu().contiguous().clone().requires_grad_(True) for pl in level]
              for level in net.deformation_net.grid.grids]
    return W, planes, net.deformation_net.grid.aabb.detach().cpu().clone()


def _seesaw_net_and_points(dev, n, seed=0):
    from mobgs_amd.synth import SynthCamera, gaussian_cloud
    net = _make_net(dev, 64, seed)
    cloud = gaussian_cloud(n, SynthCamera(), 1)
    lo, hi = cloud["xyz"].min(0).values, cloud["xyz"].max(0).values
    net.deformation_net.set_aabb((hi - 0.02 * (hi - lo)).tolist(), (lo + 0.02 * (hi - lo)).tolist())  # some outside
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        for pl in net.deformation_net.grid.planes():
            pl.copy_(0.5 + 0.5 * torch.rand(tuple(pl.shape), generator=g))
        for p in net.deformation_net.get_mlp_parameters():
            p.mul_(2.0)
    return net, cloud


def test_deform_seesaw_planes_100k_points_match_oracle(hip_device):
    """Full plane resolution of the seesaw config (35 MB of planes), 100 000 points, one time stamp."""
    from oracle import deform_torch as D
    n = 100_000
    net, cloud = _seesaw_net_and_points(hip_device, n)
    assert sum(p.numel() for p in net.deformation_net.grid.planes()) == 3 * 32 * (64 ** 2 + 128 ** 2 + 256 ** 2) \
        + 3 * 32 * 12 * (64 + 128 + 256)
    times = torch.full((n, 1), 11.0 / 23.0)
    g = torch.Generator().manual_seed(9)
    cot = [torch.randn(n, k, generator=g) for k in (3, 3, 4)]
    leaves_cpu = [cloud[k].clone().requires_grad_(True) for k in ("xyz", "scaling", "rotation")]
    W, planes, aabb = _oracle_copy(net)
    ref = D.deform_forward(*leaves_cpu, times, aabb, planes, W)
    torch.autograd.backward(ref, cot)
    leaves = [cloud[k].to(hip_device).requires_grad_(True) for k in ("xyz", "scaling", "rotation")]
    out = net(*leaves, times.to(hip_device))
    torch.autograd.backward(out, [c.to(hip_device) for c in cot])
    for a, b, name in zip(out, ref, ("pts", "scales", "rots")):
        close(a, b, 2e-5, 5e-5 * max(1.0, float(b.detach().abs().max())), name)  # fp32 sums of 96 / 128 terms, two orders
    for a, b, name in zip(leaves, leaves_cpu, ("pts", "scales", "rots")):
        sc = float(b.grad.abs().max())
        # ReLU flips and bilinear-cell flips (a coordinate within rounding of a grid line takes the neighbour's slope:
        # the sample is continuous there, its derivative is not) change a point's gradient by its own magnitude
        close_point_rows(a.grad, b.grad, 1e-3, 1e-4 * sc, "grad " + name, flip_rows=1e-3, share=1.0)
    for k, w in _weights_of(net).items():
        sc = float(W[k].grad.abs().max())
        close(w.grad, W[k].grad, 2e-3, 2e-4 * sc, "grad " + k, flip_frac=0.05, flip_atol=0.02 * sc)  # observed 3e-2 / 9e-3
    for li, level in enumerate(net.deformation_net.grid.grids):
        for pi, pl in enumerate(level):
            r = planes[li][pi].grad
            sc = float(r.abs().max())
            # observed: 6e-3 of the entries, the worst 7.6e-2 of the plane's maximum (one point's 4 x 32 taps in a plane
            # cell that few points share)
            close(pl.grad, r, 2e-3, 2e-4 * sc, f"grad plane {li}.{pi}", flip_frac=0.01, flip_atol=0.2 * sc)


def test_config3_deformed_dynamic_splats_through_the_rasterizer(hip_device):
    """200 000 static + 100 000 dynamic splats whose position / scale / rotation come out of deform_network, rendered
    at 1352x1014 (RGB+ED, 9 colour channels): image and the gradients that reach the deformation network."""
    from mobgs_amd.rendering import rasterization
    from mobgs_amd.synth import SynthCamera, splat_inputs
    from oracle import deform_torch as D
    from oracle import gsplat_cpu as Cc
    W_, H_ = 1352, 1014
    cam = SynthCamera()
    ns, nd = 200_000, 100_000
    stat = splat_inputs(ns, cam, 0, 9)
    net, cloud = _seesaw_net_and_points(hip_device, nd, seed=3)
    dyn = splat_inputs(nd, cam, 1, 9)  # same cloud (seed 1) as `cloud`: colours / opacities of the dynamic set
    times = torch.full((nd, 1), 11.0 / 23.0)
    g = torch.Generator().manual_seed(100)
    v_img = torch.randn(1, H_, W_, 10, generator=g)

    def compose(d_pts, d_scl, d_rot, to):
        means = torch.cat([to(stat["means"]), d_pts])
        scales = torch.cat([to(stat["scales"]), torch.exp(d_scl)])
        quats = torch.cat([to(stat["quats"]), d_rot])
        opac = torch.cat([to(stat["opacities"]), to(dyn["opacities"])])
        cols = torch.cat([to(stat["colors"]), to(dyn["colors"])])
        return means, quats, scales, opac, cols

    # ---- HIP: deformation -> rasterization -> backward
    dev = hip_device
    leaves = [cloud[k].to(dev).requires_grad_(True) for k in ("xyz", "scaling", "rotation")]
    d_hip = net(*leaves, times.to(dev))
    m, q, s, o, c = compose(*d_hip, lambda t: t.to(dev))
    img, alpha, meta = rasterization(m, q, s, o, c, stat["viewmats"].to(dev), stat["Ks"].to(dev), W_, H_,
                                     packed=False, backgrounds=torch.zeros(1, 9, device=dev), render_mode="RGB+ED")
    (img * v_img.to(dev)).sum().backward()
    # ---- oracle chain on the host.  The rasterizer oracle is fed the HIP network's outputs (a 1e-6 difference in a
    # position moves a quarter of the pixels by more than the image tolerance), the deformation oracle then
    # back-propagates the rasterizer oracle's cotangents through its own graph.
    W, planes, aabb = _oracle_copy(net)
    leaves_cpu = [cloud[k].clone().requires_grad_(True) for k in ("xyz", "scaling", "rotation")]
    d_cpu = D.deform_forward(*leaves_cpu, times, aabb, planes, W)
    for a_, b_, name in zip(d_hip, d_cpu, ("pts", "scales", "rots")):
        close(a_, b_, 2e-5, 5e-5 * max(1.0, float(b_.detach().abs().max())), name)
    m_c, q_c, s_c, _, _ = compose(*d_cpu, lambda t: t)
    r = Cc.rasterization_fwd_bwd(m.detach().cpu().numpy(), q.detach().cpu().numpy(), s.detach().cpu().numpy(),
                                 o.cpu().numpy(), c.cpu().numpy(), stat["viewmats"].numpy(), stat["Ks"].numpy(), W_,
                                 H_, backgrounds=np.zeros((1, 9), np.float32), render_mode="RGB+ED",
                                 v_render=v_img.numpy())
    torch.autograd.backward([m_c, q_c, s_c], [torch.from_numpy(r[k]) for k in ("v_means", "v_quats", "v_scales")])
    ref = torch.from_numpy(r["render"])
    scale = float(ref.abs().max())
    vis_depth = meta["depths"][meta["radii"] > 0]
    close_image_with_blend_flips(img, ref, r["alphas"], float(c.abs().max()),
                                 float(vis_depth.max() - vis_depth.min()), 3e-5 * scale, "image", flip_frac=2e-4,
                                 n_colour_channels=9)  # derived one-blend-step bound (was scale / 50); observed 1.5e-5
    target = ref[..., :9] / scale + 0.05 * torch.randn(ref[..., :9].shape, generator=g)
    assert abs(psnr(img[..., :9].cpu() / scale, target) - psnr(ref[..., :9] / scale, target)) <= 1e-4
    for a, b, name in zip(leaves, leaves_cpu, ("pts", "scales", "rots")):
        sc = float(b.grad.abs().max())
        close_point_rows(a.grad, b.grad, 2e-3, 1e-4 * sc, "grad " + name, flip_rows=2e-4, share=1.0)  # observed 2e-5
    for k, w in _weights_of(net).items():
        sc = float(W[k].grad.abs().max())
        close(w.grad, W[k].grad, 5e-3, 5e-3 * sc, "grad " + k, flip_frac=5e-3, flip_atol=0.01 * sc)  # observed: none
    for li, level in enumerate(net.deformation_net.grid.grids):
        for pi, pl in enumerate(level):
            b = planes[li][pi].grad
            sc = float(b.abs().max())
            close(pl.grad, b, 5e-3, 5e-4 * sc, f"grad plane {li}.{pi}", flip_frac=1e-3, flip_atol=0.02 * sc)  # observed: none
