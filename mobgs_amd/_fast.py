"""Loader of the host fast path (csrc/fastpath.cpp -> _mobgs_fast.so): the bodies of the hot autograd nodes in C++.

`get()` returns the bound module, or None when it is disabled (MOBGS_FASTPATH=0) or cannot be loaded -- the Python
bodies in ops.py / rendering.py then run instead; both end in the same C-ABI calls of libmobgs_hip.so.
"""
from __future__ import annotations

import ctypes
import importlib.util
import os
from typing import Optional

from . import _lib
from .build import FAST_PATH, build_fastpath, fastpath_is_stale

_SYMBOLS = ("mobgs_abi_version", "mobgs_project_and_bin_fused", "mobgs_prep_project_and_bin_fused", "mobgs_decoder_fwd_channels", "mobgs_decoder_bwd_channels", "mobgs_project_prep_bwd_fused", "mobgs_fused_seg_keys_len", "mobgs_last_error", "mobgs_record_stride", "mobgs_prep_fwd_many", "mobgs_prep_fwd_many_f16",
            "mobgs_prep_bwd_many", "mobgs_prep_bwd_many_f16", "mobgs_raster_fwd", "mobgs_raster_fwd_decode", "mobgs_raster_bwd", "mobgs_raster_bwd_decode", "mobgs_raster_bwd_decode_scratch_floats", "mobgs_raster_bwd_decode_finish", "mobgs_raster_bwd_reduce_decode", "mobgs_raster_bwd_reduce",
            "mobgs_decoder_fwd_many", "mobgs_decoder_bwd_many", "mobgs_decoder_bwd_blocks", "mobgs_project_bwd",
            "mobgs_project_bwd_ex",
            "mobgs_project_bwd_scratch_floats", "mobgs_project_and_bin_speculative", "mobgs_tile_order_len",
            "mobgs_keep_scan_len", "mobgs_isect_scratch_bytes", "mobgs_raster_channels_supported")
_mod = None
_tried = False
enabled = os.environ.get("MOBGS_FASTPATH", "1") != "0"
load_error: Optional[str] = None


def get():
    """The bound fast-path module or None."""
    global _mod, _tried, load_error
    if not enabled:
        return None
    if _tried:
        return _mod
    _tried = True
    try:
        if fastpath_is_stale():
            build_fastpath()
        import torch  # noqa: F401  (libtorch must be loaded before the extension)
        spec = importlib.util.spec_from_file_location("_mobgs_fast", str(FAST_PATH))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        lib = _lib.load()
        mod.bind({n: ctypes.cast(getattr(lib, n), ctypes.c_void_p).value for n in _SYMBOLS})
        _mod = mod
    except Exception as exc:  # noqa: BLE001  -- the Python bodies are a complete implementation
        load_error = f"{type(exc).__name__}: {exc}"
        _mod = None
        import warnings
        warnings.warn("mobgs_amd: the C++ host fast path could not be loaded (" + load_error + "); the Python bodies "
                      "of the autograd nodes run instead (same kernels, ~0.15 ms more host time per render). "
                      "MOBGS_FASTPATH=0 silences this.", RuntimeWarning, stacklevel=2)
    return _mod


def reset(flag: Optional[bool] = None) -> None:
    """Tests: switch the fast path on / off (None: re-read the environment)."""
    global enabled
    enabled = (os.environ.get("MOBGS_FASTPATH", "1") != "0") if flag is None else bool(flag)
