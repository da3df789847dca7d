"""Host-side helpers that need no GPU: the lazy result dict of render(), the identity-keyed cache of derived
constants, the PLY header reader on a foreign (ascii, mixed types) file."""
import os

import numpy as np
import pytest
import torch


def test_render_result_defers_groups_independently():
    from mobgs_amd.gaussian_renderer import RenderResult
    calls = []

    def aux():
        calls.append("aux")
        return {"s_render": 1, "d_render": 2}

    def vis():
        calls.append("vis")
        return {"visibility_filter": 3}

    out = RenderResult({"render": 0, "s_render": None, "d_render": None})
    out.defer(["s_render", "d_render"], aux)
    out.defer(["visibility_filter"], vis)
    assert out["render"] == 0 and calls == []            # eager entries do not trigger anything
    assert "s_render" in out and "visibility_filter" in out
    assert out["visibility_filter"] == 3 and calls == ["vis"]   # only its own group
    assert out["d_render"] == 2 and out["s_render"] == 1 and calls == ["vis", "aux"]
    assert out.get("s_render") == 1 and out.get("missing", 7) == 7
    # whole-dict views materialise everything that is still pending
    out2 = RenderResult({"render": 0})
    out2.defer(["a"], lambda: {"a": 5})
    out2.defer(["b"], lambda: {"b": 6})
    assert dict(out2.items()) == {"render": 0, "a": 5, "b": 6}
    assert sorted(out2.copy().keys()) == ["a", "b", "render"]


def test_render_result_never_leaks_its_pending_marker():
    """dict(out), {**out}, pickling, copy, pop, setdefault: all see materialised values."""
    import copy
    import pickle
    from mobgs_amd.gaussian_renderer import RenderResult, _PENDING

    def fresh():
        out = RenderResult({"render": 0})
        out.defer(["a", "b"], lambda: {"a": 5, "b": 6})
        out.defer(["c"], lambda: {"c": 7})
        return out

    want = {"render": 0, "a": 5, "b": 6, "c": 7}
    assert dict(fresh()) == want
    assert {**fresh()} == want
    assert ({"z": 1} | fresh()) == {"z": 1, **want}
    assert pickle.loads(pickle.dumps(fresh())) == want
    assert copy.copy(fresh()) == want and copy.deepcopy(fresh()) == want
    out = fresh()
    assert out.pop("a") == 5 and out["b"] == 6 and "a" not in out
    assert fresh().setdefault("c", 99) == 7 and fresh().setdefault("new", 1) == 1
    assert list(fresh()) == ["render", "a", "b", "c"] and len(fresh()) == 4
    assert all(v is not _PENDING for v in fresh().values())
    assert all(v is not _PENDING for _, v in fresh().items())


def test_derived_cache_is_keyed_on_identity_and_version():
    from mobgs_amd._lib import DerivedCache
    cache = DerivedCache()
    built = []

    def build_from(t):
        def b():
            built.append(1)
            return t * 2
        return b

    a = torch.ones(3)
    v1 = cache.get((a,), build_from(a))
    v2 = cache.get((a,), build_from(a))
    assert v1 is v2 and len(built) == 1                   # same object, unmodified -> hit
    a.add_(1)                                             # in-place change bumps the version -> rebuild
    v3 = cache.get((a,), build_from(a))
    assert len(built) == 2 and torch.equal(v3, torch.full((3,), 4.0))
    b = a.clone()                                         # equal content, different object -> rebuild
    cache.get((b,), build_from(b))
    assert len(built) == 3
    g = torch.ones(3, requires_grad=True)                 # sources that require grad are never cached
    r1 = cache.get((g,), build_from(g))
    r2 = cache.get((g,), build_from(g))
    assert r1 is not r2 and r1.requires_grad and len(built) == 5


def test_ply_reader_handles_ascii_and_mixed_property_types(tmp_path):
    from mobgs_amd import ply_io
    path = tmp_path / "a.ply"
    path.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\n"
                    "property double y\nproperty uchar red\nend_header\n1.5 2.5 7\n-1 0.25 255\n")
    names, rows = ply_io.read_ply(str(path))
    assert names == ["x", "y", "red"]
    assert np.allclose(rows, [[1.5, 2.5, 7], [-1, 0.25, 255]])
    # binary with mixed types and a second element that must be skipped
    path2 = tmp_path / "b.ply"
    dt = np.dtype([("x", "<f4"), ("n", "<i4"), ("c", "u1")])
    data = np.array([(0.5, -3, 9), (2.0, 4, 250)], dtype=dt)
    with open(path2, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nproperty int n\n"
                b"property uchar c\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n")
        f.write(data.tobytes())
    names, rows = ply_io.read_ply(str(path2))
    assert names == ["x", "n", "c"] and np.allclose(rows, [[0.5, -3, 9], [2.0, 4, 250]])


@pytest.mark.skipif(not os.path.isdir("/root/reference/scene"), reason="reference tree not present")
def test_init_geometry_helpers_match_reference_live():
    """points_from_DRTK / inverse_warp_rt1_rt2 (train.py:101,113 import them from scene.deformation) against the
    reference's own functions, run in place."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import ref_harness as RH
    import torch
    from mobgs_amd import deformation as D
    dm = RH.ref_import("scene.deformation")
    g = torch.Generator().manual_seed(0)
    B, H, W = 2, 24, 32
    depth = 1.0 + 3.0 * torch.rand(B, 1, H, W, generator=g)
    img = torch.rand(B, 3, H, W, generator=g)
    K = torch.tensor([[30.0, 0, W / 2], [0, 30.0, H / 2], [0, 0, 1]]).expand(B, 3, 3).contiguous()

    def pose(seed):
        q = torch.randn(B, 3, generator=torch.Generator().manual_seed(seed)) * 0.05
        R = torch.matrix_exp(torch.stack([torch.stack([torch.zeros(()), -v[2], v[1], v[2], torch.zeros(()), -v[0], -v[1],
                                                       v[0], torch.zeros(())]).reshape(3, 3) for v in q]))
        t = torch.randn(B, 3, 1, generator=torch.Generator().manual_seed(seed + 1)) * 0.1
        return torch.cat([R, t], dim=2)

    w1, w2 = pose(1), pose(5)
    with RH.CudaToCpu():
        ref_pts = dm.points_from_DRTK(depth, w1, K)
        ref_img, ref_grid = dm.inverse_warp_rt1_rt2(img, depth, w1, w2, K, torch.inverse(K), ret_grid=True)
    assert torch.allclose(D.points_from_DRTK(depth, w1, K), ref_pts, atol=1e-5)
    out, grid = D.inverse_warp_rt1_rt2(img, depth, w1, w2, K, torch.inverse(K), ret_grid=True)
    assert torch.allclose(grid, ref_grid, atol=1e-5) and torch.allclose(out, ref_img, atol=1e-5)
    assert float((ref_grid == 2).float().mean()) > 0.0  # some pixels do leave the image


def test_round5_host_switches_and_keys():
    """Host logic added in round 5 that needs no GPU: the per-node zero-cotangent gate context, per-call tuning copies, the
    signature that keys the implicit mid-exposure cache of get_flow(), row-order bookkeeping of the model classes."""
    import mobgs_amd.gaussian_renderer as G
    import mobgs_amd.rendering as R
    from mobgs_amd import _lib
    from mobgs_amd.camera import PinholeCamera
    from mobgs_amd.gaussian_model import GaussianParams
    from mobgs_amd.synth import SynthCamera, dynamic_extras, gaussian_cloud
    # gate: nested contexts restore the previous state; nodes read the state at construction time
    assert R._zero_gate[0] == 0
    with R.zero_cotangent_gate():
        assert R._zero_gate[0] == 1
        with R.zero_cotangent_gate(False):
            assert R._zero_gate[0] == 0
        assert R._zero_gate[0] == 1
    assert R._zero_gate[0] == 0
    # tuning copies: every field carried over, overrides applied, the module's struct untouched
    t = R.tuning.copy(gate_zero_cotangent=1, coherent_order=1)
    for name, _ in _lib.MobgsTuning._fields_:
        if name not in ("gate_zero_cotangent", "coherent_order"):
            assert getattr(t, name) == getattr(R.tuning, name)
    assert (t.gate_zero_cotangent, t.coherent_order) == (1, 1)
    assert (R.tuning.gate_zero_cotangent, R.tuning.coherent_order) == (0, 0)
    g1 = R._tuning_gated()
    assert g1 is R._tuning_gated() and g1.gate_zero_cotangent == 1   # cached until a field of `tuning` changes
    old = R.tuning.heavy_tile_len
    try:
        R.tuning.heavy_tile_len = 77
        g2 = R._tuning_gated()
        assert g2 is not g1 and g2.heavy_tile_len == 77
    finally:
        R.tuning.heavy_tile_len = old
    # cache signature: an in-place update (what an optimiser step does) and a new tensor both change it; reading does not
    scam = SynthCamera().scaled(64, 48)
    sp, dp = gaussian_cloud(50, scam, 0), gaussian_cloud(20, scam, 1)
    stat = GaussianParams(sp, None, None, "cpu", requires_grad=True)
    dyn = GaussianParams(dp, dynamic_extras(dp["xyz"], 0), stat.rgbdecoder, "cpu", requires_grad=True)
    cam = PinholeCamera(64, 48, scam.K, torch.eye(4), scam.time, scam.max_time, device="cpu")
    sig = lambda: G._mid_signature(cam, stat, dyn)
    s0 = sig()
    assert sig() == s0
    with torch.no_grad():
        dyn.control_xyz.add_(0.5)
    s1 = sig()
    assert s1 != s0
    stat._opacity = stat._opacity.detach().clone().requires_grad_(True)
    assert sig() != s1
    s2 = sig()
    with torch.no_grad():
        assert sig() != s2   # (grad mode is part of the key)
    # row order: a fresh model is not "coherent"; spatial_sort_ needs the library only for nothing -- it is torch code
    assert stat.rows_coherent == -1 and not G._rows_coherent(stat)
    xyz = stat._xyz.detach().clone()
    order = stat.spatial_sort_()
    assert stat.rows_coherent == 50 and G._rows_coherent(stat) and torch.equal(stat._xyz.detach(), xyz[order])
    assert sorted(order.tolist()) == list(range(50)) and stat._xyz.requires_grad
    o2 = dyn.spatial_sort_()
    assert dyn.rows_coherent == 20 and dyn.control_xyz.shape == (20, 12, 3) and sorted(o2.tolist()) == list(range(20))


def test_flow_head_outputs_are_used_the_way_the_reference_uses_them():
    """gaussian_renderer._FlowHead hands the maps of a get_flow() group out as views of one autograd node: the
    reference's own use -- torch.cat of the calls' maps, THEN the in-place normalisation of /root/reference/train.py:658-660
    -- works and back-propagates; an in-place edit of a returned map itself is PyTorch's multiple-views error (documented in
    get_flow's docstring)."""
    import pytest
    import torch
    from mobgs_amd.gaussian_renderer import _FlowHead
    a = torch.rand(1, 4, 5, 2, requires_grad=True)
    b = torch.rand(1, 4, 5, 2, requires_grad=True)
    o = _FlowHead.apply(a * 1.0, b * 1.0)
    cat = torch.cat([o[0], o[1]], 0).unsqueeze(0)           # train.py:583-586
    cat[..., 0] = cat[..., 0] / 4                            # :659 (W - 1)
    cat[..., 1] = cat[..., 1] / 3                            # :660 (H - 1)
    (2.0 * cat - 1.0).sum().backward()
    assert torch.allclose(a.grad[..., 0], torch.full_like(a.grad[..., 0], 0.5))
    assert torch.allclose(b.grad[..., 1], torch.full_like(b.grad[..., 1], 2.0 / 3.0))
    o2 = _FlowHead.apply(a * 1.0, b * 1.0)
    with pytest.raises(RuntimeError, match="multiple views|view"):
        o2[0][..., 0] = o2[0][..., 0] / 2


def test_hints_are_kept_per_scene_and_per_thread():
    """rendering.hint_scope / scene_token (round 6): the workload key of arena capacities, list-length
    hints and key-segment strides carries a token of the (static, dynamic) pair being rendered -- two scenes of equal size do
    not share an entry, the token of a pair is stable, and the scope is thread-local."""
    import threading
    import torch
    import mobgs_amd.rendering as R

    class PC:
        pass
    s1, d1, s2, d2 = PC(), PC(), PC(), PC()
    t1, t2 = R.scene_token(s1, d1), R.scene_token(s2, d2)
    assert t1 != t2 and t1 == R.scene_token(s1, d1) and R.scene_token(s2, d1) not in (t1, t2)
    dev = torch.device("cpu")
    with R.hint_scope(t1):
        k1 = R._workload_key(dev, 1, 300000, 1352, 1014)
        seen = {}

        def other():
            seen["outside"] = R._workload_key(dev, 1, 300000, 1352, 1014)
            with R.hint_scope(t2):
                seen["k2"] = R._workload_key(dev, 1, 300000, 1352, 1014)
        th = threading.Thread(target=other)
        th.start()
        th.join()
        assert R._workload_key(dev, 1, 300000, 1352, 1014) == k1    # the other thread's scope did not leak into this one
    assert k1[0] == t1 and seen["k2"][0] == t2 and seen["outside"][0] == 0 and k1[1:] == seen["k2"][1:]
    assert R._workload_key(dev, 1, 300000, 1352, 1014)[0] == 0      # outside any scope: the operator API's shared entry
