"""ctypes binding of libmobgs_hip.so (the C ABI declared in include/mobgs_hip.h).

The product path has NO CPU fallback: if the library cannot be loaded, or a tensor is not on a HIP device,
the call raises.  PyTorch is used only for device memory and the current stream.
"""
from __future__ import annotations

import ctypes
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p
from typing import Optional

import torch

from .build import LIB_PATH, build_extension, is_stale

_lib: Optional[ctypes.CDLL] = None


class MobgsTuning(ctypes.Structure):
    """include/mobgs_hip.h MobgsTuning: per-call policy (-1 = library default).  The library keeps no state; a
    caller-side instance (mobgs_amd.rendering.tuning) is passed by pointer with every call that consults it."""
    _fields_ = [("heavy_tile_len", ctypes.c_int32), ("longest_list_hint", ctypes.c_int32),
                ("quadrant_culling", ctypes.c_int32), ("block_walk", ctypes.c_int32),
                ("bwd_block_walk", ctypes.c_int32), ("geometry_per_camera", ctypes.c_int32),
                ("bwd_mfma", ctypes.c_int32), ("gate_zero_cotangent", ctypes.c_int32),
                ("coherent_order", ctypes.c_int32), ("static_rows", ctypes.c_int32), ("cover_slots", ctypes.c_int32)]

    def __init__(self, heavy_tile_len=-1, longest_list_hint=-1, quadrant_culling=-1, block_walk=-1, bwd_block_walk=-1,
                 geometry_per_camera=0, bwd_mfma=-1, gate_zero_cotangent=0, coherent_order=0, static_rows=0,
                 cover_slots=0):
        super().__init__(heavy_tile_len, longest_list_hint, quadrant_culling, block_walk, bwd_block_walk,
                         geometry_per_camera, bwd_mfma, gate_zero_cotangent, coherent_order, static_rows, cover_slots)

    def copy(self, **overrides):
        """A per-call copy with some fields replaced."""
        t = MobgsTuning(*[getattr(self, n) for n, _ in self._fields_])
        for k, v in overrides.items():
            setattr(t, k, v)
        return t

    def ref(self):
        return ctypes.cast(ctypes.pointer(self), c_void_p)

    def address(self) -> int:
        return ctypes.addressof(self)

class MobgsPrepInputs(ctypes.Structure):
    """include/mobgs_hip.h MobgsPrepInputs: the raw parameters of the two sets (device pointers)."""
    _fields_ = [("Ns", ctypes.c_int32), ("Nd", ctypes.c_int32)] + [(n, c_void_p) for n in (
        "times", "s_xyz", "s_scaling", "s_rotation", "s_opacity", "s_fdc", "s_ft", "d_control", "d_ncp", "d_scaling",
        "d_rotation", "d_omega", "d_opacity", "d_fdc", "d_ft", "d_trbf")]


P = c_void_p
ABI_VERSION = 9  # include/mobgs_hip.h MOBGS_ABI_VERSION
_SIGS = {
    "mobgs_version": (c_char_p, []),
    "mobgs_abi_version": (c_int, []),
    "mobgs_last_error": (c_char_p, []),
    "mobgs_record_stride": (c_int, [c_int]),
    "mobgs_raster_channels_supported": (c_int, [c_int]),
    "mobgs_raster_path": (c_int, [c_int, c_int, c_int, P]),
    "mobgs_project_fwd": (c_int, [c_int, c_int, P, P, P, P, P, c_int, c_int, c_float, c_float, c_float, c_float,
                                  P, P, P, P, P, P]),
    "mobgs_project_bwd_scratch_floats": (c_size_t, [c_int, c_int]),
    "mobgs_project_bwd": (c_int, [c_int, c_int, P, P, P, P, P, c_int, c_int, c_float, P, P, P, P, P, P, P, P, P,
                                  P, P]),
    "mobgs_project_bwd_ex": (c_int, [c_int, c_int, c_int, P, P, P, P, P, c_int, c_int, c_float, P, P, P, P, P, P, P,
                                     P, P, P, P]),
    "mobgs_isect_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mobgs_keep_scan_len": (c_size_t, [c_int]),
    "mobgs_tile_order_len": (c_size_t, [c_int]),
    "mobgs_isect_offsets": (c_int, [c_int] * 8 + [P] * 5 + [c_int] + [P] * 4 + [c_int64] + [P] * 2 + [P, P]),
    "mobgs_isect_emit_sort": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int64, c_int64] + [P] * 7 + [P]),
    "mobgs_raster_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P, c_int, P, P, P, P, P, P, P,
                                 P, P, P, P, P, P]),
    "mobgs_raster_fwd_decode": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P, c_int, P, P, P, P, P, P, P,
                                        P, P, P, P, P, c_int, P, c_int, P, P, P, P, P, P]),
    "mobgs_raster_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int] + [P] * 16 + [P, P]),
    "mobgs_raster_bwd_decode": (c_int, [c_int, c_int, c_int, c_int] + [P] * 15 + [c_int, P, c_int] + [P] * 6 + [P, P]),
    "mobgs_raster_bwd_decode_finish": (c_int, [c_int, c_int, c_int, P, c_int, P, P, P, c_int, c_int, P]),
    "mobgs_raster_bwd_reduce_decode": (c_int, [c_int, c_int] + [P] * 11 + [c_int, c_int, P, c_int, P, P, P, c_int, c_int, P]),
    "mobgs_raster_bwd_decode_scratch_floats": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "mobgs_cotangent_probe": (c_int, [c_int, P, P, P, P]),
    "mobgs_raster_bwd_reduce": (c_int, [c_int, c_int, c_int, c_int] + [P] * 11 + [P]),
    "mobgs_project_and_bin": (c_int, [c_int, c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, c_float,
                                      c_float, c_int, P, P, P, P, P, P, P, P, P, c_int, P, P, c_int64, P, P, P, P, P,
                                      P]),
    "mobgs_isect_emit_sort_speculative": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int64, c_int64] + [P] * 8 + [P]),
    "mobgs_project_and_bin_speculative": (c_int, [c_int, c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_float,
                                                  c_float, c_float, c_float, c_int, P, P, P, P, P, P, P, P, P, c_int,
                                                  P, P, c_int64, P, P, P, c_int64, P, c_int64, P, c_int, c_int, P, P,
                                                  P]),
    "mobgs_fused_seg_keys_len": (c_size_t, [c_int, c_int]),
    "mobgs_fused_max_seg_stride": (c_int, []),
    "mobgs_project_and_bin_fused": (c_int, [c_int, c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, c_float,
                                            c_float, c_int, P, P, P, P, P, P, P, P, P, c_int, P, P, c_int64, P, P, c_int,
                                            P, P, c_int64, P, c_int64, P, c_int, c_int, P, P, P]),
    "mobgs_prep_project_and_bin_fused": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_float, c_float, c_float, c_float,
                                                 c_int, P, P, P, P, P, P, P, P, P, c_int, P, P, c_int64, P, P, c_int, P,
                                                 P, c_int64, P, c_int64, P, P, P]),
    "mobgs_project_prep_bwd_fused": (c_int, [c_int, P, P, P, P, P, c_int, c_int, c_float] + [P] * 10 + [c_int, c_int] +
                                     [P] * 7 + [c_int, P]),
    "mobgs_densify_stats": (c_int, [c_int, P, c_int, P, P, P, P, P, P]),
    "mobgs_densify_select": (c_int, [c_int, c_int, P, P, P, c_float, c_float, P, P, P]),
    "mobgs_mask_indices": (c_int, [c_int, P, c_int, P, P, P]),
    "mobgs_rows_gather": (c_int, [c_int, P, P, P, P, P, c_int, c_int, P]),
    "mobgs_split_children": (c_int, [c_int, c_int, c_int, P, P, P, P, P]),
    "mobgs_normals_fwd": (c_int, [c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float, P, P, P]),
    "mobgs_normals_bwd": (c_int, [c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float, P, P, P, P]),
    "mobgs_raster_class_fwd": (c_int, [c_int] * 7 + [P] * 9 + [P, P]),
    "mobgs_raster_class_bwd": (c_int, [c_int] * 7 + [P] * 15 + [P, P]),
    "mobgs_pack_records": (c_int, [c_int, c_int, c_int, P, P, P, c_int, P, c_int, P, P, P, P]),
    "mobgs_raster_layers_fwd": (c_int, [c_int] * 7 + [P] * 8 + [P]),
    "mobgs_raster_layers_bwd": (c_int, [c_int] * 8 + [P] * 21 + [P]),  # incl. 7 host pointer arrays of length 3
    "mobgs_prep_fwd": (c_int, [c_int, c_int] + [P] * 21 + [P]),
    "mobgs_prep_bwd": (c_int, [c_int, c_int] + [P] * 23 + [c_int, P]),
    "mobgs_prep_fwd_f16": (c_int, [c_int, c_int] + [P] * 21 + [P]),
    "mobgs_prep_bwd_f16": (c_int, [c_int, c_int] + [P] * 23 + [c_int, P]),
    "mobgs_prep_fwd_many": (c_int, [c_int, c_int, c_int] + [P] * 21 + [P]),
    "mobgs_prep_bwd_many": (c_int, [c_int, c_int, c_int] + [P] * 23 + [c_int, P]),
    "mobgs_prep_fwd_many_f16": (c_int, [c_int, c_int, c_int] + [P] * 21 + [P]),
    "mobgs_prep_bwd_many_f16": (c_int, [c_int, c_int, c_int] + [P] * 23 + [c_int, P]),
    "mobgs_decoder_fwd": (c_int, [c_int, c_int, c_int, c_int] + [P] * 9 + [P]),
    "mobgs_decoder_bwd_blocks": (c_int, [c_int]),
    "mobgs_decoder_bwd": (c_int, [c_int, c_int, c_int, c_int] + [P] * 16 + [c_int, c_int, P]),
    "mobgs_decoder_fwd_many": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, c_int64, P, c_int, P, c_int, P, P, P, P,
                                       P]),
    "mobgs_decoder_bwd_many": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, c_int64, P, c_int, P, c_int] + [P] * 11
                               + [c_int, c_int, P]),
    "mobgs_decoder_fwd_channels": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, c_int64, P, c_int, P, c_int, P, P, P,
                                           P, P, c_int, c_int, P]),
    "mobgs_decoder_bwd_channels": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, c_int64, P, c_int, P, c_int] +
                                   [P] * 11 + [c_int, c_int, P, c_int, c_int, P]),
    "mobgs_ssim_l1_blocks": (c_int, [c_int, c_int, c_int]),
    "mobgs_ssim_l1_fwd": (c_int, [c_int, c_int, c_int, P, P, P, P, P]),
    "mobgs_ssim_l1_bwd": (c_int, [c_int, c_int, c_int, P, P, P, P, P, P]),
    "mobgs_flow_warp_loss_blocks": (c_int, [c_int, c_int, c_int]),
    "mobgs_flow_warp_loss_fwd": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P]),
    "mobgs_flow_warp_loss_bwd": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, P]),
    "mobgs_flow_warp_loss_bwd_scratch_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mobgs_hexplane_fwd": (c_int, [c_int] + [P] * 7 + [P]),
    "mobgs_hexplane_bwd": (c_int, [c_int] + [P] * 11 + [P]),
    "mobgs_hexplane_bwd_scratch_bytes": (c_size_t, [c_int, P, P]),
    "mobgs_deform_mlp_fwd": (c_int, [c_int] + [P] * 14 + [P]),
    "mobgs_blce_saved_floats": (c_size_t, []),
    "mobgs_blce_fwd": (c_int, [P, c_int, c_int, P, P, P, P, P, P]),
    "mobgs_adam_step": (c_int, [c_int, P, ctypes.c_double, ctypes.c_double, ctypes.c_double, P]),
    "mobgs_blce_bwd": (c_int, [P, P, c_int, c_int, P, P, P, P, P]),
    "mobgs_deform_mlp_bwd_blocks": (c_int, [c_int]),
    "mobgs_deform_mlp_grad_floats": (c_size_t, []),
    "mobgs_deform_mlp_bwd": (c_int, [c_int] + [P] * 20 + [P]),
}
# entry points added by later translation units (bound if present in the header AND the library)
_OPTIONAL_SIGS = {}


def register_optional(name: str, restype, argtypes) -> None:
    _OPTIONAL_SIGS[name] = (restype, argtypes)
    if _lib is not None:
        _bind(_lib, name, restype, argtypes)


def _bind(lib, name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    """Load (building first if the .so is missing or older than its sources and hipcc is available)."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing and is_stale():
        try:
            build_extension()
        except Exception as exc:  # noqa: BLE001
            if not LIB_PATH.exists():
                raise RuntimeError(
                    f"libmobgs_hip.so is missing and could not be built ({exc}); the mobgs_amd product path "
                    "has no CPU fallback") from exc
    if not LIB_PATH.exists():
        raise RuntimeError(f"{LIB_PATH} not found; run `python -m mobgs_amd.build`")
    lib = ctypes.CDLL(str(LIB_PATH))
    # include/mobgs_hip.h MOBGS_ABI_VERSION these bindings were written against: a stale or foreign build of the
    # library (MOBGS_LIB) with other signatures / scratch formats must not be driven with shifted arguments
    got = lib.mobgs_abi_version() if hasattr(lib, "mobgs_abi_version") else 0
    if got != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: mobgs_abi_version() = {got}, these bindings need {ABI_VERSION} "
                           "(rebuild with `python -m mobgs_amd.build`)")
    for name, (restype, argtypes) in {**_SIGS, **_OPTIONAL_SIGS}.items():
        if name in _SIGS or hasattr(lib, name):
            _bind(lib, name, restype, argtypes)
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().mobgs_last_error().decode()
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("mobgs_amd: tensors must live on a HIP device (device='cuda'); there is no CPU path")
    if not t.is_contiguous():
        raise RuntimeError("mobgs_amd: internal error, non-contiguous tensor passed to the C ABI")
    return c_void_p(t.data_ptr())


class DerivedCache:
    """One-entry memo for a small tensor derived from other tensors (a padded background row, packed camera
    parameters ...): rebuilt unless the SAME tensor objects, unmodified since (Tensor._version), are passed again.
    The sources are kept alive by the entry, so a recycled address can never alias them.  Sources that require grad
    are never cached (the derived tensor must stay in their autograd graph)."""

    def __init__(self):
        self.srcs, self.versions, self.value = (), (), None

    def get(self, srcs, build):
        if any(t.requires_grad for t in srcs):
            return build()
        if len(srcs) == len(self.srcs) and all(a is b for a, b in zip(srcs, self.srcs)) and \
                all(t._version == v for t, v in zip(srcs, self.versions)):
            return self.value
        value = build()
        self.srcs, self.versions, self.value = tuple(srcs), tuple(t._version for t in srcs), value
        return value


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """The current HIP stream of the current device as a void*.  (torch.cuda.current_stream() builds a Stream object
    through several Python layers, ~7 us a call -- 17 calls per render step made it 0.1 ms of a host-bound step.)"""
    return c_void_p(stream_int())


def stream_int() -> int:
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def f32c(t: torch.Tensor) -> torch.Tensor:
    """float32 + contiguous (no copy when already so)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def attr_c(t: torch.Tensor, half: bool) -> torch.Tensor:
    """A per-splat attribute array in the storage type the kernel was chosen for: contiguous, and float16 when
    `half` (no copy when it already is -- the fp16-storage path never widens in HBM), float32 otherwise."""
    want = torch.float16 if half else torch.float32
    if t.dtype != want:
        global attr_conversions
        attr_conversions += 1
        t = t.to(want)
    return t.contiguous()


attr_conversions = 0  # dtype conversions of attribute arrays on the way into a kernel (tests assert none happen)
