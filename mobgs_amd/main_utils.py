"""`get_normals` of the reference's main_utils.py (:95-141), on the GPU.

    pred_normal = get_normals(pred_depth + 1e-6, camera_metadata)      # train.py:590, once per view and iteration

`camera_metadata` is anything with the attributes the reference reads from its dycheck camera: principal_point_x/y,
scale_factor_x/y, skew, and (optionally) use_center (default True: pixel centres at +0.5, dycheck_geometry/
camera.py:600-613).  Returns [1,3,H,W]; differentiable w.r.t. the depth map.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, f32c, ptr, stream


class _Normals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, cam):
        zc = f32c(z)
        H, W = zc.shape[-2:]
        out = torch.empty(3, H, W, dtype=torch.float32, device=zc.device)
        check(_lib.load().mobgs_normals_fwd(H, W, *cam, ptr(zc), ptr(out), stream()), "mobgs_normals_fwd")
        ctx.save_for_backward(zc)
        ctx.cam = cam
        ctx.shape = z.shape
        return out

    @staticmethod
    def backward(ctx, g):
        (zc,) = ctx.saved_tensors
        H, W = zc.shape[-2:]
        v_z = torch.empty(H, W, dtype=torch.float32, device=zc.device)
        check(_lib.load().mobgs_normals_bwd(H, W, *ctx.cam, ptr(zc), ptr(f32c(g)), ptr(v_z), stream()),
              "mobgs_normals_bwd")
        return v_z.reshape(ctx.shape), None


def get_normals(z: torch.Tensor, camera_metadata) -> torch.Tensor:
    """z: depth [1,H,W] (or [H,W]).  -> unit normals [1,3,H,W], zero on the 1-pixel border."""
    m = camera_metadata
    offset = 0.5 if getattr(m, "use_center", True) else 0.0
    cam = (float(m.scale_factor_x), float(m.scale_factor_y), float(m.principal_point_x),
           float(m.principal_point_y), float(getattr(m, "skew", 0.0)), offset)
    return _Normals.apply(z, cam)[None]
