"""Densification / optimiser surgery on the GPU (mobgs_amd.densify + csrc/densify.hip) against the states the
reference's GaussianModel produced for the same inputs (tests/golden/densify.npz, make_golden.py:gen_densify)."""
import numpy as np
import pytest
import torch

from helpers import load

pytestmark = pytest.mark.gpu

GROUPS = ["xyz", "control_xyz", "current_control_num", "f_dc", "f_rest", "f_t", "opacity", "scaling", "rotation",
          "omega", "zeta", "trbf_center", "trbf_scale", "motion"]
AUX = ["xyz_gradient_accum", "denom", "max_radii2D", "_deformation_table", "_deformation_accum"]


class Opt:
    percent_dense = 0.01
    position_lr_init = 0.00016
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


def _model(fx, tag, dev):
    from mobgs_amd.densify import TrainableGaussians
    t = lambda k: torch.from_numpy(fx[f"{tag}.{k}"])  # noqa: E731
    params = {"xyz": t("xyz"), "scaling": t("scaling"), "rotation": t("rotation"), "opacity": t("opacity"),
              "features_dc": t("f_dc"), "features_t": t("f_t")}
    dyn = {"omega": t("omega"), "trbf_center": t("trbf_center"), "control_xyz": t("control_xyz"),
           "current_control_num": t("current_control_num"), "f_rest": t("f_rest"), "zeta": t("zeta"),
           "trbf_scale": t("trbf_scale"), "motion": t("motion"), "_deformation_table": t("_deformation_table")}
    pc = TrainableGaussians(params, dyn, device=dev)
    pc.training_setup(Opt())
    # the Adam state the reference had after its first step
    for gr in pc.optimizer.param_groups:
        g = gr["name"]
        if f"{tag}.{g}.exp_avg" in fx:
            pc.optimizer.state[gr["params"][0]] = {
                "step": torch.tensor(1.0), "exp_avg": t(f"{g}.exp_avg").to(dev).clone(),
                "exp_avg_sq": t(f"{g}.exp_avg_sq").to(dev).clone()}
    for a in ("xyz_gradient_accum", "denom", "max_radii2D", "_deformation_accum"):
        getattr(pc, a).copy_(t(a).to(dev))
    return pc


def _check(pc, fx, tag, loose=()):
    st = pc.table_state()
    n = fx[f"{tag}.xyz"].shape[0]
    assert pc.get_xyz.shape[0] == n
    for g in GROUPS:
        for kind in ("", ".exp_avg", ".exp_avg_sq"):
            key = f"{tag}.{g}{kind}"
            if key not in fx:
                continue
            ref = torch.from_numpy(fx[key])
            got = st[g + kind].cpu()
            assert got.shape == ref.shape, (key, got.shape, ref.shape)
            if g + kind in loose:
                assert torch.allclose(got, ref, rtol=2e-6, atol=2e-6), (key, float((got - ref).abs().max()))
            else:
                assert torch.equal(got, ref), key
    for a in AUX:
        ref = torch.from_numpy(fx[f"{tag}.{a}"])
        got = st[a].cpu()
        assert got.shape == ref.shape, (a, got.shape, ref.shape)
        if a == "xyz_gradient_accum":  # sqrt(gx^2 + gy^2): 1 ulp between libm / device sqrt paths
            assert torch.allclose(got, ref, rtol=3e-7, atol=0), f"{tag}.{a}"
        else:
            assert torch.equal(got, ref), f"{tag}.{a}"
    # the attributes are the optimiser's parameters, and the moments are the table's
    groups = {gr["name"]: gr for gr in pc.optimizer.param_groups}
    for g, attr in (("xyz", "_xyz"), ("opacity", "_opacity"), ("control_xyz", "control_xyz")):
        assert groups[g]["params"][0] is getattr(pc, attr)
        assert pc.optimizer.state[getattr(pc, attr)]["exp_avg"].data_ptr() == st[g + ".exp_avg"].data_ptr()


def _stats(pc, fx, dev):
    for it in range(2):
        pc.add_densification_stats(torch.from_numpy(fx[f"stats{it}.viewspace_grad"]).to(dev),
                                   torch.from_numpy(fx[f"stats{it}.visible"]).to(dev),
                                   radii=torch.from_numpy(fx[f"stats{it}.radii"]).to(dev))


def test_densify_sequence_matches_reference(hip_device):
    fx = load("densify")
    dev = hip_device
    pc = _model(fx, "s0", dev)
    _stats(pc, fx, dev)
    _check(pc, fx, "s1")
    # continue from the reference's own statistics so that later stages are compared bit for bit
    pc.xyz_gradient_accum.copy_(torch.from_numpy(fx["s1.xyz_gradient_accum"]).to(dev))
    grads = torch.from_numpy(fx["grads"]).to(dev)
    thr, extent = float(fx["max_grad"]), float(fx["extent"])
    pc.densify_and_clone(grads, thr, extent)
    _check(pc, fx, "s2")
    pc.densify_and_splitv2(grads, thr, extent, 2, samples=torch.from_numpy(fx["split.samples"]).to(dev))
    _check(pc, fx, "s3", loose=("xyz", "scaling"))
    # re-synchronise the two fp32-rounded fields, then prune and reset bit-exactly
    st = pc.table_state()
    st["xyz"].copy_(torch.from_numpy(fx["s3.xyz"]).to(dev))
    st["scaling"].copy_(torch.from_numpy(fx["s3.scaling"]).to(dev))
    pc.prune_points(torch.from_numpy(fx["prune.mask"]).to(dev))
    _check(pc, fx, "s4")
    pc.reset_opacity()
    _check(pc, fx, "s5", loose=("opacity",))


def test_fused_pruneclone_equals_clone_then_split(hip_device):
    """densify_pruneclone (one selection pass, one gather) == densify_and_clone + densify_and_splitv2."""
    fx = load("densify")
    dev = hip_device
    pc = _model(fx, "s1", dev)
    pc.densify_pruneclone(float(fx["max_grad"]), 0.005, float(fx["extent"]), None, 2,
                          samples=torch.from_numpy(fx["split.samples"]).to(dev))
    _check(pc, fx, "s3", loose=("xyz", "scaling"))


def test_optimizer_steps_through_resizes(hip_device):
    """Adam keeps working on the re-keyed parameters: a step after clone+split changes exactly the live rows, new
    rows start from zero moments, and the table grows past its first capacity."""
    fx = load("densify")
    dev = hip_device
    pc = _model(fx, "s1", dev)
    for _ in range(4):  # 600 -> several thousand rows: crosses the initial capacity
        pc.xyz_gradient_accum.fill_(1.0)
        pc.denom.fill_(1.0)
        n0 = pc.get_xyz.shape[0]
        pc.densify_pruneclone(0.5, 0.005, float(fx["extent"]), None, 2)
        n1 = pc.get_xyz.shape[0]
        assert n1 > n0
        before = pc._xyz.detach().clone()
        for gr in pc.optimizer.param_groups:
            for p in gr["params"]:
                if p.requires_grad and p.numel():
                    p.grad = torch.ones_like(p)
        pc.optimizer.step()
        assert pc._xyz.shape[0] == n1 and not torch.equal(pc._xyz.detach(), before)
        st = pc.optimizer.state[pc._xyz]
        assert st["exp_avg"].shape[0] == n1 and bool(torch.isfinite(st["exp_avg"]).all())
    assert pc.get_xyz.shape[0] > 2500
