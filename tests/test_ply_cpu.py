"""PLY checkpoint layout against what the reference's own save_ply hands to plyfile (tests/golden/densify.npz:
ply.names / ply.rows were captured by running GaussianModel.save_ply with a recording stand-in for plyfile)."""
import numpy as np
import torch

from helpers import load


class _PC:
    pass


def _container(fx, tag="s0"):
    pc = _PC()
    t = lambda k: torch.from_numpy(fx[f"{tag}.{k}"])  # noqa: E731
    pc._xyz, pc._scaling, pc._rotation, pc._opacity = t("xyz"), t("scaling"), t("rotation"), t("opacity")
    pc._features_dc, pc._features_rest, pc._features_t = t("f_dc"), t("f_rest"), t("f_t")
    pc._omega, pc._zeta, pc._trbf_center, pc._trbf_scale = t("omega"), t("zeta"), t("trbf_center"), t("trbf_scale")
    pc._motion, pc.control_xyz, pc.current_control_num = t("motion"), t("control_xyz"), t("current_control_num")
    pc.rgbdecoder = torch.nn.Linear(2, 2)
    return pc


def test_ply_layout_matches_reference_save_ply(tmp_path):
    from mobgs_amd import ply_io
    fx = load("densify")
    pc = _container(fx)
    assert ply_io.attribute_names(pc) == [str(s) for s in fx["ply.names"]]
    assert np.array_equal(ply_io.attribute_rows(pc), fx["ply.rows"])
    path = str(tmp_path / "point_cloud" / "iteration_1" / "point_cloud.ply")
    ply_io.save_ply(pc, path)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", f"element vertex {fx['ply.rows'].shape[0]}"]
    assert lines[3:-1] == [f"property float {n}" for n in fx["ply.names"]]
    assert body == fx["ply.rows"].astype("<f4").tobytes()
    assert set(torch.load(path.replace(".ply", ".pt")).keys()) == {"weight", "bias"}


def test_ply_round_trip(tmp_path):
    from mobgs_amd import ply_io
    fx = load("densify")
    pc = _container(fx, "s3")
    path = str(tmp_path / "pc.ply")
    ply_io.save_ply(pc, path)
    params, dyn = ply_io.load_ply(path)
    assert torch.equal(params["xyz"], pc._xyz) and torch.equal(params["scaling"], pc._scaling)
    assert torch.equal(params["rotation"], pc._rotation) and torch.equal(params["opacity"], pc._opacity)
    assert torch.equal(params["features_dc"], pc._features_dc) and torch.equal(params["features_t"], pc._features_t)
    assert torch.equal(dyn["control_xyz"], pc.control_xyz) and torch.equal(dyn["omega"], pc._omega)
    assert torch.equal(dyn["current_control_num"], pc.current_control_num)
    assert torch.equal(dyn["motion"], pc._motion) and torch.equal(dyn["trbf_center"], pc._trbf_center)
    assert dyn["f_rest"].shape == pc._features_rest.shape
