"""Read-side Gaussian container.

`GaussianParams` exposes the attributes/properties of the reference's `GaussianModel` that render()/get_flow()
read (/root/reference/scene/gaussian_model.py:91-106 activations, :209-254 accessors; parameter shapes from
create_from_pcd* :406-582).  Optimiser surgery, densification and PLY I/O stay with the caller (out of scope).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .helper_model import Sandwich


class GaussianParams:
    def __init__(self, params: Dict[str, torch.Tensor], dynamic: Optional[Dict[str, torch.Tensor]] = None,
                 decoder: Optional[torch.nn.Module] = None, device="cpu", requires_grad: bool = False,
                 attr_dtype: torch.dtype = torch.float32):
        """attr_dtype=torch.float16: the per-splat ATTRIBUTES (scaling, rotation, omega, opacity, features) are kept
        in half precision in HBM (BASELINE config #5; the render kernels widen them in registers).  Positions, spline
        control points and time centres always stay fp32."""
        def P(t, grad=True, attr=False):
            t = t.detach().clone().to(device)
            if attr and t.is_floating_point():
                t = t.to(attr_dtype)
            if grad and requires_grad and t.is_floating_point():
                t.requires_grad_(True)
            return t

        self.attr_dtype = attr_dtype
        self.is_dynamic = dynamic is not None
        self.rows_coherent = -1   # the row count at which spatial_sort_() last ordered the rows (-1: never)
        self._xyz = P(params["xyz"])
        self._scaling = P(params["scaling"], attr=True)
        self._rotation = P(params["rotation"], attr=True)
        self._opacity = P(params["opacity"], attr=True)
        self._features_dc = P(params["features_dc"], attr=True)
        self._features_t = P(params["features_t"], attr=True)
        n = self._xyz.shape[0]
        dyn = dynamic or {}
        self._omega = P(dyn.get("omega", torch.zeros(n, 4)), attr=True)
        self._trbf_center = P(dyn.get("trbf_center", torch.zeros(n, 1)))
        self.control_xyz = P(dyn.get("control_xyz", (params["xyz"] * 100.0)[:, None, :].repeat(1, 12, 1)))
        self.current_control_num = P(dyn.get("current_control_num", torch.full((n, 1), 12, dtype=torch.int64)),
                                     grad=False)
        self.rgbdecoder = decoder if decoder is not None else Sandwich(9, 3).to(device)
        self.scaling_activation = torch.exp
        self.opacity_activation = torch.sigmoid
        self.rotation_activation = F.normalize

    # --- storage order ------------------------------------------------------------------------------
    PER_SPLAT = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_t", "_omega", "_trbf_center",
                 "control_xyz", "current_control_num")

    def sort_positions(self) -> torch.Tensor:
        """[n,3] representative positions of the rows: the static position, or the mean spline control point."""
        return self.control_xyz.detach().float().mean(1) if self.is_dynamic else self._xyz.detach().float()

    @torch.no_grad()
    def spatial_sort_(self) -> torch.Tensor:
        """Store the rows along a Morton curve of their positions (rendering.spatial_order) -> the permutation applied.
        The renderer's binning kernel works on 2048 consecutive bounding-box intersections at a time; when consecutive
        rows are neighbours in space those fall on a handful of tiles and are ranked in LDS with one global atomic per
        (workgroup, tile) -- 1352x1014 / 300 k splats: bin 49.6 -> 29.9 us -- and every kernel that walks the rows
        streams (gaussian_renderer passes rendering.COHERENT while `rows_coherent` equals the row count; otherwise it
        falls back to a cached enumeration ORDER over unsorted rows, which costs an indirection in three kernels).
        Rendering does not depend on the row order (ties in depth aside); optimiser state kept outside this object must
        be permuted by the caller with the returned indices (densify.TrainableGaussians does it for its own).  Call it
        after loading and after densification; positions drift slowly in between."""
        from .rendering import spatial_order
        order = spatial_order(self.sort_positions()).long()
        for name in self.PER_SPLAT:
            t = getattr(self, name, None)
            if torch.is_tensor(t) and t.shape[:1] == order.shape:
                t.data = t.data[order]
                t.grad = None
                m = getattr(t, "master", None)
                if m is not None:
                    m.data = m.data[order]
                    m.grad = None
        from . import gaussian_renderer as _GR
        _GR.parameters_changed()   # (.data moved in place: caches keyed on Tensor._version must be told)
        self.rows_coherent = int(order.numel())
        return order

    # --- accessors with the reference's names -------------------------------------------------------
    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation_stat(self):
        return F.normalize(self._rotation)

    def get_rotation_dy(self, rotation, delta_t):
        return rotation + delta_t * self._omega

    @property
    def get_control_xyz(self):
        return self.control_xyz

    @property
    def get_trbfcenter(self):
        return self._trbf_center

    def get_features(self, deltat):
        return torch.cat((self._features_dc, deltat * self._features_t), dim=1)

    @property
    def get_features_static(self):
        return torch.cat((self._features_dc, 0.0 * self._features_t), dim=1)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def leaf_tensors(self, dynamic: bool) -> Dict[str, torch.Tensor]:
        names = ["_scaling", "_rotation", "_opacity", "_features_dc", "_features_t"]
        names += ["control_xyz", "_omega"] if dynamic else ["_xyz"]
        return {k: getattr(self, k) for k in names}

    # ---- training with half-precision attribute storage (BASELINE config #5) ---------------------------------------
    def enable_fp32_masters(self, dynamic: bool) -> Dict[str, torch.Tensor]:
        """For attr_dtype=float16 sets that are TRAINED: every half-stored leaf gets an fp32 master copy (`leaf.master`,
        a leaf tensor requiring grad).  The render kernels keep reading the halves; inside an ops.LeafGradSink the prep
        backward accumulates the attribute gradients in fp32 straight into `master.grad` (no half gradient exists, so
        nothing saturates at 65504 and a cross-rank SUM runs in fp32: distributed.FlatGradients over trainable_tensors());
        the optimiser steps the masters (fp32 Adam moments) and sync_half() rounds them into the stored halves -- the
        usual master-weight scheme, with the fp16 copy being what the kernels stream.  -> trainable_tensors(dynamic)."""
        for k, t in self.leaf_tensors(dynamic).items():
            if t.dtype == torch.float16 and getattr(t, "master", None) is None:
                t.master = t.detach().float().requires_grad_(True)
        return self.trainable_tensors(dynamic)

    def trainable_tensors(self, dynamic: bool) -> Dict[str, torch.Tensor]:
        """What an optimiser / a flat gradient buffer should hold: the fp32 master of a half-stored leaf, else the leaf."""
        return {k: (getattr(t, "master", None) if getattr(t, "master", None) is not None else t)
                for k, t in self.leaf_tensors(dynamic).items()}

    @torch.no_grad()
    def sync_half(self, dynamic: bool) -> None:
        """After optimizer.step(): round the masters into the stored halves (one multi-tensor copy)."""
        pairs = [(t, t.master) for t in self.leaf_tensors(dynamic).values() if getattr(t, "master", None) is not None]
        if pairs:
            torch._foreach_copy_([a for a, _ in pairs], [b for _, b in pairs])
