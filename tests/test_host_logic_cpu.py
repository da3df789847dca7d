"""Host-side helpers that need no GPU: the lazy result dict of render(), the identity-keyed cache of derived
constants, the PLY header reader on a foreign (ascii, mixed types) file."""
import numpy as np
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
