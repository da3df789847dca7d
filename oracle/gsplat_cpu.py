"""ctypes wrapper of oracle/gsplat_cpu.c (CPU ORACLE -- test infrastructure, NOT product code).

PARITY UNPINNED vs. real gsplat (see the header of gsplat_cpu.c).  numpy in / numpy out, float32.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_float, c_int, c_int64, c_void_p
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = HERE / "gsplat_cpu.c"
LIB = HERE / "_build" / "libgsplat_cpu.so"
_lib = None


def build(force: bool = False) -> Path:
    if force or not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        LIB.parent.mkdir(exist_ok=True)
        cmd = ["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", str(SRC), "-o", str(LIB), "-lm"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"gcc failed:\n{r.stdout}")
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(LIB))
        _lib.ora_isect_count.restype = c_int64
        _lib.ora_num_threads.restype = c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def num_threads() -> int:
    return int(lib().ora_num_threads())


def set_num_threads(n: int) -> None:
    lib().ora_set_num_threads(c_int(n))


def project_fwd(means, quats, scales, viewmats, Ks, W, H, eps2d=0.3, near=0.01, far=1e10, radius_clip=0.0):
    means, quats, scales, viewmats, Ks = map(_f, (means, quats, scales, viewmats, Ks))
    C, N = viewmats.shape[0], means.shape[0]
    radii = np.zeros((C, N), np.int32)
    means2d = np.zeros((C, N, 2), np.float32)
    depths = np.zeros((C, N), np.float32)
    conics = np.zeros((C, N, 3), np.float32)
    lib().ora_project_fwd(c_int(C), c_int(N), _p(means), _p(quats), _p(scales), _p(viewmats), _p(Ks), c_int(W),
                          c_int(H), c_float(eps2d), c_float(near), c_float(far), c_float(radius_clip), _p(radii),
                          _p(means2d), _p(depths), _p(conics))
    return radii, means2d, depths, conics


def project_bwd(means, quats, scales, viewmats, Ks, W, H, radii, conics, v_means2d, v_depths, v_conics, eps2d=0.3):
    means, quats, scales, viewmats, Ks, conics = map(_f, (means, quats, scales, viewmats, Ks, conics))
    v_means2d, v_depths, v_conics = map(_f, (v_means2d, v_depths, v_conics))
    C, N = viewmats.shape[0], means.shape[0]
    v_means = np.zeros((N, 3), np.float32)
    v_quats = np.zeros((N, 4), np.float32)
    v_scales = np.zeros((N, 3), np.float32)
    v_viewmats = np.zeros((C, 4, 4), np.float32)
    radii = np.ascontiguousarray(radii, dtype=np.int32)
    lib().ora_project_bwd(c_int(C), c_int(N), _p(means), _p(quats), _p(scales), _p(viewmats), _p(Ks), c_int(W),
                          c_int(H), c_float(eps2d), _p(radii), _p(conics), _p(v_means2d), _p(v_depths),
                          _p(v_conics), _p(v_means), _p(v_quats), _p(v_scales), _p(v_viewmats))
    return v_means, v_quats, v_scales, v_viewmats


def isect(means2d, radii, depths, W, H):
    """-> tiles_per_gauss [C,N], isect_ids i64 [I], flatten_ids i32 [I], isect_offsets i32 [C,th,tw]"""
    means2d, depths = _f(means2d), _f(depths)
    radii = np.ascontiguousarray(radii, dtype=np.int32)
    C, N = radii.shape
    tw, th = (W + 15) // 16, (H + 15) // 16
    tpg = np.zeros((C, N), np.int32)
    n = int(lib().ora_isect_count(c_int(C), c_int(N), c_int(W), c_int(H), _p(means2d), _p(radii), _p(tpg)))
    ids = np.zeros(max(n, 1), np.int64)
    flat = np.zeros(max(n, 1), np.int32)
    offs = np.zeros((C, th, tw), np.int32)
    lib().ora_isect_sort(c_int(C), c_int(N), c_int(W), c_int(H), _p(means2d), _p(radii), _p(depths), c_int64(n),
                         _p(ids), _p(flat), _p(offs))
    return tpg, ids[:n], flat[:n], offs


def raster_fwd(means2d, conics, colors, opacities, backgrounds, offs, flat, W, H):
    """colors [C,N,D], opacities [C,N] -> render [C,H,W,D], alphas [C,H,W], last_ids [C,H,W]"""
    means2d, conics, colors, opacities, backgrounds = map(_f, (means2d, conics, colors, opacities, backgrounds))
    C, N, D = colors.shape
    assert D <= 64
    render = np.zeros((C, H, W, D), np.float32)
    alphas = np.zeros((C, H, W), np.float32)
    last = np.zeros((C, H, W), np.int32)
    offs = np.ascontiguousarray(offs, np.int32)
    flat = np.ascontiguousarray(flat, np.int32)
    lib().ora_raster_fwd(c_int(C), c_int(N), c_int(D), c_int(W), c_int(H), _p(means2d), _p(conics), _p(colors),
                         _p(opacities), _p(backgrounds), _p(offs), _p(flat), c_int64(flat.shape[0]), _p(render),
                         _p(alphas), _p(last))
    return render, alphas, last


def raster_bwd(means2d, conics, colors, opacities, backgrounds, offs, flat, W, H, alphas, last, v_render, v_alphas):
    means2d, conics, colors, opacities, backgrounds = map(_f, (means2d, conics, colors, opacities, backgrounds))
    alphas, v_render, v_alphas = map(_f, (alphas, v_render, v_alphas))
    C, N, D = colors.shape
    v_means2d = np.zeros((C, N, 2), np.float32)
    v_conics = np.zeros((C, N, 3), np.float32)
    v_colors = np.zeros((C, N, D), np.float32)
    v_opac = np.zeros((C, N), np.float32)
    offs = np.ascontiguousarray(offs, np.int32)
    flat = np.ascontiguousarray(flat, np.int32)
    last = np.ascontiguousarray(last, np.int32)
    lib().ora_raster_bwd(c_int(C), c_int(N), c_int(D), c_int(W), c_int(H), _p(means2d), _p(conics), _p(colors),
                         _p(opacities), _p(backgrounds), _p(offs), _p(flat), c_int64(flat.shape[0]), _p(alphas),
                         _p(last), _p(v_render), _p(v_alphas), _p(v_means2d), _p(v_conics), _p(v_colors), _p(v_opac))
    return v_means2d, v_conics, v_colors, v_opac


def rasterization_fwd_bwd(means, quats, scales, opacities, colors, viewmats, Ks, W, H, backgrounds=None,
                          render_mode="RGB", v_render=None, v_alphas=None):
    """Whole operator like gsplat.rasterization (C>=1, unpacked).  With cotangents v_render [C,H,W,X] /
    v_alphas [C,H,W] also runs the backward pass and returns the input gradients."""
    means, quats, scales, opacities, colors, viewmats, Ks = map(_f, (means, quats, scales, opacities, colors,
                                                                     viewmats, Ks))
    C, N = viewmats.shape[0], means.shape[0]
    radii, means2d, depths, conics = project_fwd(means, quats, scales, viewmats, Ks, W, H)
    tpg, ids, flat, offs = isect(means2d, radii, depths, W, H)
    cols = np.broadcast_to(colors, (C, N, colors.shape[-1])) if colors.ndim == 2 else colors
    bg = _f(backgrounds)
    with_depth = render_mode in ("RGB+D", "RGB+ED")
    if with_depth:
        cols = np.concatenate([cols, depths[..., None]], -1)
        if bg is not None:
            bg = np.concatenate([bg, np.zeros((C, 1), np.float32)], -1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None]
        if bg is not None:
            bg = np.zeros((C, 1), np.float32)
    cols = np.ascontiguousarray(cols, np.float32)
    opac = np.ascontiguousarray(np.broadcast_to(opacities, (C, N)), np.float32)
    render, alphas, last = raster_fwd(means2d, conics, cols, opac, bg, offs, flat, W, H)
    out = render.copy()
    ed = render_mode in ("ED", "RGB+ED")
    if ed:
        out[..., -1] = render[..., -1] / np.maximum(alphas, 1e-10)
    res = {"render": out, "alphas": alphas, "radii": radii, "means2d": means2d, "depths": depths, "conics": conics,
           "tiles_per_gauss": tpg, "isect_ids": ids, "flatten_ids": flat, "isect_offsets": offs, "last_ids": last}
    if v_render is None:
        return res
    v_render = _f(v_render).copy()
    v_alphas = np.zeros((C, H, W), np.float32) if v_alphas is None else _f(v_alphas).copy()
    if ed:  # out_d = acc_d / max(alpha, 1e-10)
        a = np.maximum(alphas, 1e-10)
        g = v_render[..., -1]
        v_alphas = v_alphas + np.where(alphas > 1e-10, -g * render[..., -1] / (a * a), 0.0).astype(np.float32)
        v_render[..., -1] = g / a
    v_means2d, v_conics, v_cols, v_opac = raster_bwd(means2d, conics, cols, opac, bg, offs, flat, W, H, alphas, last,
                                                     v_render, v_alphas)
    v_depths = None
    if with_depth:
        v_depths = v_cols[..., -1]
        v_cols = v_cols[..., :-1]
    elif render_mode in ("D", "ED"):
        v_depths = v_cols[..., 0]
        v_cols = None
    v_means, v_quats, v_scales, v_viewmats = project_bwd(means, quats, scales, viewmats, Ks, W, H, radii, conics,
                                                         v_means2d, v_depths, v_conics)
    res.update({"v_means": v_means, "v_quats": v_quats, "v_scales": v_scales, "v_viewmats": v_viewmats,
                "v_opacities": v_opac.sum(0) if opacities.ndim == 1 else v_opac,
                "v_colors": None if v_cols is None else (v_cols.sum(0) if colors.ndim == 2 else v_cols),
                "v_means2d": v_means2d})
    return res


# ---------------------------------------------------------------------------------------------------------------------
# The same C routines behind torch.autograd on the CPU: a drop-in for the `rasterization=` argument of
# oracle/render_torch.render() / get_flow(), so that a FULL-SIZE render() -- prep and Sandwich decoder in plain torch
# (render_torch), projection / lists / compositing forward and backward in C -- finishes in seconds on the host
# (tests/test_gpu_fullsize.py: the render()-level oracle of the benchmark's own kernel selection).
# Test infrastructure like everything in this file.
# ---------------------------------------------------------------------------------------------------------------------
def torch_rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, packed=False,
                        backgrounds=None, render_mode="RGB", **_unused):
    """gsplat.rendering.rasterization's signature as the reference calls it -> (colors [C,H,W,X], alphas [C,H,W,1],
    {"radii", "means2d" (autograd non-leaf)}); CPU tensors, differentiable w.r.t. means / quats / scales / opacities /
    colors / viewmats."""
    import torch

    W, H = int(width), int(height)

    class _Proj(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means, quats, scales, viewmats):
            a = [t.detach().numpy() for t in (means, quats, scales, viewmats)]
            radii, means2d, depths, conics = project_fwd(a[0], a[1], a[2], a[3], Ks.detach().numpy(), W, H)
            ctx.np = (a, radii, conics)
            ctx.mark_non_differentiable(*(r := [torch.from_numpy(radii)]))
            return r[0], torch.from_numpy(means2d), torch.from_numpy(depths), torch.from_numpy(conics)

        @staticmethod
        def backward(ctx, _vr, v_m2d, v_dep, v_con):
            a, radii, conics = ctx.np
            C, N = radii.shape
            z = lambda g, shape: np.zeros(shape, np.float32) if g is None else g.numpy()  # noqa: E731
            vm, vq, vs, vv = project_bwd(a[0], a[1], a[2], a[3], Ks.detach().numpy(), W, H, radii, conics,
                                         z(v_m2d, (C, N, 2)), z(v_dep, (C, N)), z(v_con, (C, N, 3)))
            return tuple(torch.from_numpy(x) for x in (vm, vq, vs, vv))

    class _Raster(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means2d, conics, cols, opac, radii, depths):
            C, N = radii.shape
            _, _, flat, offs = isect(means2d.detach().numpy(), radii.numpy(), depths.detach().numpy(), W, H)
            bg = None if backgrounds is None else backgrounds.detach().numpy()
            if bg is not None and bg.shape[-1] < cols.shape[-1]:
                bg = np.concatenate([bg, np.zeros((C, cols.shape[-1] - bg.shape[-1]), np.float32)], -1)
            a = [np.ascontiguousarray(t.detach().numpy(), np.float32) for t in (means2d, conics, cols, opac)]
            render, alphas, last = raster_fwd(a[0], a[1], a[2], a[3], bg, offs, flat, W, H)
            ctx.np = (a, bg, offs, flat, alphas, last)
            return torch.from_numpy(render), torch.from_numpy(alphas)

        @staticmethod
        def backward(ctx, v_render, v_alphas):
            a, bg, offs, flat, alphas, last = ctx.np
            C, _, D = a[2].shape
            vr = np.zeros((C, H, W, D), np.float32) if v_render is None else v_render.contiguous().numpy()
            va = np.zeros((C, H, W), np.float32) if v_alphas is None else v_alphas.contiguous().numpy()
            g = raster_bwd(a[0], a[1], a[2], a[3], bg, offs, flat, W, H, alphas, last, vr, va)
            return tuple(torch.from_numpy(x) for x in g) + (None, None)

    C, N = viewmats.shape[0], means.shape[0]
    radii, means2d, depths, conics = _Proj.apply(means, quats, scales, viewmats)
    cols = colors.expand(C, N, colors.shape[-1]) if colors.dim() == 2 else colors
    with_depth = render_mode in ("RGB+D", "RGB+ED")
    if with_depth:
        cols = torch.cat([cols, depths[..., None]], -1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None]
    opac = opacities.expand(C, N) if opacities.dim() == 1 else opacities
    render, alphas = _Raster.apply(means2d, conics, cols.contiguous(), opac.contiguous(), radii, depths)
    if render_mode in ("ED", "RGB+ED"):
        render = torch.cat([render[..., :-1], render[..., -1:] / alphas[..., None].clamp(min=1e-10)], -1)
    return render, alphas[..., None], {"radii": radii, "means2d": means2d, "depths": depths, "conics": conics}
