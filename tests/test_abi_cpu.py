"""CPU checks of the drop-in boundary: the C-ABI library loads here (no GPU), exports every symbol that
include/mobgs_hip.h declares, and the host-side wrappers refuse what they do not implement / CPU tensors."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mobgs_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mobgs_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from mobgs_amd import _lib, build
    path = build.build_extension()
    assert os.path.exists(path)
    lib = ctypes.CDLL(str(path))
    syms = declared_symbols()
    assert len(syms) >= 18
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    h = _lib.load()
    assert h.mobgs_version().decode().startswith("mobgs_hip")
    assert h.mobgs_record_stride(10) == 16 and h.mobgs_record_stride(1) == 8 and h.mobgs_record_stride(2) == 8
    assert h.mobgs_raster_channels_supported(10) == 1 and h.mobgs_raster_channels_supported(7) == 0
    # every ctypes signature in the binding refers to an exported symbol
    for name in _lib._SIGS:
        assert name in syms, f"{name} bound in _lib.py but not declared in the header"


def test_size_queries_need_no_gpu():
    from mobgs_amd import _lib
    h = _lib.load()
    assert h.mobgs_project_bwd_scratch_floats(1, 300000) == ((300000 + 255) // 256) * 16
    assert h.mobgs_isect_scratch_bytes(300000, 5440, 1 << 22) > 4 * (5440 + 3 * (1 << 22))  # counters + 3 ints per slot
    assert h.mobgs_decoder_bwd_blocks(1352 * 1014) >= 256


def test_bad_arguments_are_refused_before_any_launch():
    """Error behaviour of the C ABI: a bad argument returns MOBGS_E_INVALID with a message, without touching a device
    (so this runs here).  mobgs_raster_bwd_reduce needs the packed records since the gradient slots carry raw sums."""
    from mobgs_amd import _lib
    h = _lib.load()
    none = ctypes.c_void_p(None)
    rc = h.mobgs_raster_bwd_reduce(1, 5, 3, 0, none, none, none, none, none, none, none, none, none, none, none, none)
    assert rc == -1 and b"mobgs_raster_bwd_reduce" in h.mobgs_last_error()
    rc = h.mobgs_raster_bwd_reduce(0, 5, 3, 0, none, none, none, none, none, none, none, none, none, none, none, none)
    assert rc == -1
    with pytest.raises(RuntimeError, match="mobgs_raster_bwd_reduce"):
        _lib.check(rc, "mobgs_raster_bwd_reduce")


def test_abi_version_is_checked():
    """The bindings refuse a library whose mobgs_abi_version() differs from the header they were written against
    (a signature changed mid-list without any version signal)."""
    from mobgs_amd import _lib
    h = _lib.load()
    text = open(HEADER).read()
    assert h.mobgs_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define MOBGS_ABI_VERSION (\d+)", text).group(1))


def test_no_cpu_fallback():
    from mobgs_amd.rendering import fully_fused_projection, rasterization
    n = 8
    a = dict(means=torch.rand(n, 3), quats=torch.rand(n, 4), scales=torch.rand(n, 3), opacities=torch.rand(n),
             colors=torch.rand(n, 3), viewmats=torch.eye(4)[None], Ks=torch.eye(3)[None], width=32, height=32)
    with pytest.raises(RuntimeError, match="HIP device"):
        rasterization(packed=False, **a)
    with pytest.raises(RuntimeError, match="HIP device"):
        fully_fused_projection(a["means"], None, a["quats"], a["scales"], a["viewmats"], a["Ks"], 32, 32)


def test_unsupported_gsplat_options_raise():
    from mobgs_amd.rendering import _pad_channels, fully_fused_projection, rasterization
    n = 4
    a = dict(means=torch.rand(n, 3), quats=torch.rand(n, 4), scales=torch.rand(n, 3), opacities=torch.rand(n),
             colors=torch.rand(n, 3), viewmats=torch.eye(4)[None], Ks=torch.eye(3)[None], width=32, height=32)
    for kw in (dict(packed=True), dict(packed=False, sh_degree=3), dict(packed=False, absgrad=True),
               dict(packed=False, rasterize_mode="antialiased"), dict(packed=False, camera_model="fisheye"),
               dict(packed=False, tile_size=8), dict(packed=False, sparse_grad=True),
               dict(packed=False, distributed=True)):
        with pytest.raises(NotImplementedError):
            rasterization(**a, **kw)
    with pytest.raises(ValueError):
        rasterization(packed=False, render_mode="XYZ", **a)
    with pytest.raises(NotImplementedError):
        fully_fused_projection(a["means"], torch.rand(n, 3, 3), None, None, a["viewmats"], a["Ks"], 32, 32)
    assert [_pad_channels(d) for d in (1, 2, 3, 4, 5, 9, 10, 11, 13, 17, 26)] == [1, 2, 3, 4, 9, 9, 10, 12, 16, 26, 26]
    with pytest.raises(NotImplementedError):
        _pad_channels(27)


def test_render_cluster_argument_raises_like_the_reference():
    from mobgs_amd.gaussian_renderer import render
    with pytest.raises(NameError):
        render(None, None, None, None, None, cluster=1)


def test_host_fast_path_builds_loads_and_binds():
    """csrc/fastpath.cpp (the autograd-node bodies in C++) compiles against the installed libtorch with g++, loads,
    and binds every C-ABI entry point it calls from the ctypes handle."""
    from mobgs_amd import _fast
    _fast.reset(True)
    try:
        m = _fast.get()
        assert m is not None, _fast.load_error
        for name in ("prep_fwd", "prep_bwd", "raster_fwd", "raster_bwd", "raster_bwd_reduce", "decoder_fwd",
                     "decoder_bwd", "project_bwd"):
            assert callable(getattr(m, name))
    finally:
        _fast.reset(None)
