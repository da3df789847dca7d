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
