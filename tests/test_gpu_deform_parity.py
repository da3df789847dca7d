"""GPU parity of the deformation API against the fixture produced by the reference's own deform_network."""
import numpy as np
import pytest
import torch

from helpers import close, load

pytestmark = pytest.mark.gpu


class _Args:
    net_width, timebase_pe, defor_depth, posebase_pe, scale_rotation_pe, opacity_pe = 128, 4, 1, 10, 2, 2
    timenet_width, timenet_output, bounds, grid_pe = 64, 32, 1.6, 0
    kplanes_config = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32,
                      "resolution": [8, 8, 8, 4]}
    multires = [1, 2, 4]
    no_dx = no_grid = no_ds = no_dr = empty_voxel = static_mlp = apply_rotation = False
    no_do = no_dshs = True


def _net_from_fixture(fx, dev):
    from mobgs_amd.deformation import deform_network
    net = deform_network(_Args()).to(dev)
    d = net.deformation_net
    T = lambda k: torch.from_numpy(fx[k]).to(dev)  # noqa: E731
    with torch.no_grad():
        d.feature_out[0].weight.copy_(T("w_w0"))
        d.feature_out[0].bias.copy_(T("w_b0"))
        for name, seq in (("pos", d.pos_deform), ("scl", d.scales_deform), ("rot", d.rotations_deform)):
            seq[1].weight.copy_(T(f"w_{name}_w1"))
            seq[1].bias.copy_(T(f"w_{name}_b1"))
            seq[3].weight.copy_(T(f"w_{name}_w2"))
            seq[3].bias.copy_(T(f"w_{name}_b2"))
        for li, level in enumerate(d.grid.grids):
            for pi, pl in enumerate(level):
                pl.copy_(T(f"plane_{li}_{pi}"))
        d.grid.aabb.copy_(T("in_aabb"))
    return net


def test_state_dict_keys_match_reference_layout(hip_device):
    from mobgs_amd.deformation import deform_network
    net = deform_network(_Args())
    keys = set(net.state_dict().keys())
    for k in ("deformation_net.grid.aabb", "deformation_net.grid.grids.0.0", "deformation_net.grid.grids.2.5",
              "deformation_net.feature_out.0.weight", "deformation_net.pos_deform.1.weight",
              "deformation_net.pos_deform.3.bias", "deformation_net.scales_deform.3.weight",
              "deformation_net.rotations_deform.1.bias", "timenet.0.weight", "timenet.2.bias", "time_poc", "pos_poc",
              "rotation_scaling_poc", "opacity_poc"):
        assert k in keys, k
    fx = load("deform")
    assert sum(p.numel() for p in net.parameters()) == int(fx["n_params"][0])


def test_deform_network_matches_reference_fixture(hip_device):
    fx = load("deform")
    net = _net_from_fixture(fx, hip_device)
    T = lambda k: torch.from_numpy(fx[k]).to(hip_device)  # noqa: E731
    pts, scales, rots = (T(k).requires_grad_(True) for k in ("in_pts", "in_scales", "in_rots"))
    o_pts, o_scl, o_rot = net(pts, scales, rots, T("in_times"))
    close(o_pts, fx["out_pts"], 2e-5, 2e-5, "pts")
    close(o_scl, fx["out_scales"], 2e-5, 2e-5, "scales")
    close(o_rot, fx["out_rots"], 2e-5, 2e-5, "rotations")
    ((o_pts * T("cot_pts")).sum() + (o_scl * T("cot_scales")).sum() + (o_rot * T("cot_rots")).sum()).backward()
    for name, t in (("pts", pts), ("scales", scales), ("rots", rots)):
        ref = fx["grad_" + name]
        close(t.grad, ref, 1e-3, 1e-4 * float(np.abs(ref).max()), f"grad {name}")
    d = net.deformation_net
    pairs = [("w0", d.feature_out[0].weight), ("b0", d.feature_out[0].bias)]
    for name, seq in (("pos", d.pos_deform), ("scl", d.scales_deform), ("rot", d.rotations_deform)):
        pairs += [(f"{name}_w1", seq[1].weight), (f"{name}_b1", seq[1].bias), (f"{name}_w2", seq[3].weight),
                  (f"{name}_b2", seq[3].bias)]
    for k, p in pairs:
        ref = fx["gw_" + k]
        close(p.grad, ref, 1e-3, 1e-4 * float(np.abs(ref).max()) + 1e-6, f"grad {k}")
    for li, level in enumerate(d.grid.grids):
        for pi, pl in enumerate(level):
            ref = fx[f"gplane_{li}_{pi}"]
            close(pl.grad, ref, 1e-3, 1e-4 * float(np.abs(ref).max()) + 1e-6, f"grad plane {li}.{pi}")
