"""CPU ORACLE No. 3 (test infrastructure, NOT product code): a float64, list-free restatement of what gsplat v1.4.0's
rasterization() computes -- written to share NOTHING with the tile machinery of the two other restatements
(oracle/gsplat_torch.py, oracle/gsplat_cpu.c) and of the HIP kernels.

PARITY UNPINNED vs. real gsplat (see oracle/gsplat_torch.py: gsplat==1.4.0 is an absent third-party dependency of the
reference, /root/reference/README.md:26, imported at /root/reference/gaussian_renderer/__init__.py:15).  What this file
adds is independence: the two existing restatements and the kernels all build per-tile lists (bounding-box binning,
64-bit keys, a sort, offsets, batches) and walk them; a mistake in that shared picture would pass every comparison
between them.  Here there are no lists, no keys, no offsets and no batches:

    for every splat with radius > 0, in ONE global order (float32 depth ascending, ties by index -- the order every
    per-tile list of SURVEY.md Appendix A.2 is a sub-sequence of):
        for every pixel at once (vectorised over the whole image):
            the pixel takes part iff its 16x16 tile lies in the splat's tile rectangle (A.2 -- this IS part of the
            semantics: a splat is never evaluated outside the tiles its 3-sigma box touches, although alpha can still
            be 0.011 > 1/255 there), then the blend rule of A.3 with its skip / stop tests.

All arithmetic in float64 on float32 inputs; projection (A.1) restated independently as well.  Gradients come from torch
autograd over this forward pass: a third, float64 gradient reference for the hand-written backward kernels.
Sizes: a loop over splats of whole-image tensor operations -- seconds at 2 k splats x 160 x 128 pixels.

Only tests/ may import this module.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

F64 = torch.float64


def project(means: Tensor, quats: Tensor, scales: Tensor, viewmat: Tensor, K: Tensor, W: int, H: int, eps2d: float = 0.3,
            near: float = 0.01, far: float = 1e10, radius_clip: float = 0.0):
    """SURVEY.md Appendix A.1 for one camera, float64.  -> radii [N] int64, means2d [N,2], depths [N], conics [N,3]
    (entries with radii == 0 hold zeros)."""
    mu, q, s = means.to(F64), quats.to(F64), scales.to(F64)
    V, Km = viewmat.to(F64), K.to(F64)
    R, t = V[:3, :3], V[:3, 3]
    p = mu @ R.T + t
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    qn = q / q.norm(dim=-1, keepdim=True)
    w_, i_, j_, k_ = qn.unbind(-1)
    Rq = torch.stack([1 - 2 * (j_ * j_ + k_ * k_), 2 * (i_ * j_ - w_ * k_), 2 * (i_ * k_ + w_ * j_),
                      2 * (i_ * j_ + w_ * k_), 1 - 2 * (i_ * i_ + k_ * k_), 2 * (j_ * k_ - w_ * i_),
                      2 * (i_ * k_ - w_ * j_), 2 * (j_ * k_ + w_ * i_), 1 - 2 * (i_ * i_ + j_ * j_)], -1).reshape(-1, 3, 3)
    M = Rq * s[:, None, :]
    cov_w = M @ M.transpose(1, 2)
    cov_c = R @ cov_w @ R.T
    fx, fy, cx, cy = Km[0, 0], Km[1, 1], Km[0, 2], Km[1, 2]
    tanx, tany = 0.5 * W / fx, 0.5 * H / fy
    lim_xp, lim_xn = (W - cx) / fx + 0.3 * tanx, cx / fx + 0.3 * tanx
    lim_yp, lim_yn = (H - cy) / fy + 0.3 * tany, cy / fy + 0.3 * tany
    zs = torch.where(z.abs() < 1e-30, torch.full_like(z, 1e-30), z)
    tx = zs * torch.minimum(torch.maximum(x / zs, -lim_xn), lim_xp)
    ty = zs * torch.minimum(torch.maximum(y / zs, -lim_yn), lim_yp)
    zero = torch.zeros_like(zs)
    J = torch.stack([fx / zs, zero, -fx * tx / (zs * zs), zero, fy / zs, -fy * ty / (zs * zs)], -1).reshape(-1, 2, 3)
    cov2 = J @ cov_c @ J.transpose(1, 2)
    a = cov2[:, 0, 0] + eps2d
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + eps2d
    det = a * c - b * b
    m2d = torch.stack([fx * x / zs + cx, fy * y / zs + cy], -1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.01))
    radius = torch.ceil(3.0 * torch.sqrt(lam))
    ok = (z >= near) & (z <= far) & (det > 0) & (radius > radius_clip)
    ok = ok & ~((m2d[:, 0] + radius <= 0) | (m2d[:, 0] - radius >= W) | (m2d[:, 1] + radius <= 0) | (m2d[:, 1] - radius >= H))
    dsafe = torch.where(det > 0, det, torch.ones_like(det))
    conics = torch.stack([c / dsafe, -b / dsafe, a / dsafe], -1)
    okf = ok.to(F64)
    radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int64)
    return radii, m2d * okf[:, None], z * okf, conics * okf[:, None]


def tile_rect(mx: float, my: float, radius: int, tile_w: int, tile_h: int, tile: int = 16) -> Tuple[int, int, int, int]:
    """A.2, evaluated in float32 like the definition (the inputs of this rule ARE float32 numbers and the rule rounds)."""
    f = np.float32
    tr = f(radius) / f(tile)
    tx, ty = f(mx) / f(tile), f(my) / f(tile)
    x0 = int(min(max(0.0, math.floor(tx - tr)), tile_w))
    x1 = int(min(max(0.0, math.ceil(tx + tr)), tile_w))
    y0 = int(min(max(0.0, math.floor(ty - tr)), tile_h))
    y1 = int(min(max(0.0, math.ceil(ty + tr)), tile_h))
    return x0, y0, x1, y1


def global_order(depths: Tensor, radii: Tensor) -> list:
    """Indices of the splats with radius > 0 by (float32 depth bits, index): the order of every tile list (A.2: the
    sort key holds the bits of the FLOAT32 depth; positive floats order like their bit patterns)."""
    d32 = depths.detach().to(torch.float32).cpu().numpy()
    idx = np.nonzero(radii.detach().cpu().numpy() > 0)[0]
    return idx[np.lexsort((idx, d32[idx]))].tolist()


def composite(means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor, depths: Tensor, radii: Tensor, W: int,
              H: int, background: Optional[Tensor] = None):
    """A.3 without lists.  means2d [N,2], conics [N,3], colors [N,D], opacities [N], depths [N], radii [N] (any float
    dtype: evaluated in float64; float32 depths / means2d / radii decide order and tile membership).
    -> image [H,W,D], alpha [H,W], last [H,W] = RANK in global_order() of the last blended splat (-1: none)."""
    m, cn, col, op = means2d.to(F64), conics.to(F64), colors.to(F64), opacities.to(F64)
    tile_w, tile_h = (W + 15) // 16, (H + 15) // 16
    ys, xs = torch.meshgrid(torch.arange(H, dtype=F64), torch.arange(W, dtype=F64), indexing="ij")
    px, py = xs + 0.5, ys + 0.5
    T = torch.ones(H, W, dtype=F64)
    done = torch.zeros(H, W, dtype=torch.bool)
    out = torch.zeros(H, W, col.shape[-1], dtype=F64)
    last = torch.full((H, W), -1, dtype=torch.int64)
    m32 = means2d.detach().to(torch.float32).cpu().numpy()
    rad = radii.detach().cpu().numpy()
    for rank, i in enumerate(global_order(depths, radii)):
        x0, y0, x1, y1 = tile_rect(m32[i, 0], m32[i, 1], int(rad[i]), tile_w, tile_h)
        if x1 <= x0 or y1 <= y0:
            continue
        # pixels of the tiles [y0, y1) x [x0, x1)
        ya, yb, xa, xb = 16 * y0, min(16 * y1, H), 16 * x0, min(16 * x1, W)
        dx, dy = m[i, 0] - px[ya:yb, xa:xb], m[i, 1] - py[ya:yb, xa:xb]
        sigma = 0.5 * (cn[i, 0] * dx * dx + cn[i, 2] * dy * dy) + cn[i, 1] * dx * dy
        alpha = torch.clamp(op[i] * torch.exp(-sigma), max=0.999)
        Tw = T[ya:yb, xa:xb]
        valid = ~done[ya:yb, xa:xb] & ~((sigma < 0) | (alpha < 1.0 / 255.0))
        Tn = Tw * (1.0 - alpha)
        stop = valid & (Tn <= 1e-4)
        blend = valid & ~stop
        wgt = torch.where(blend, alpha * Tw, torch.zeros_like(Tw))
        # (functional updates of the window: autograd-safe)
        out = torch.cat([out[:ya], torch.cat([out[ya:yb, :xa], out[ya:yb, xa:xb] + wgt[..., None] * col[i],
                                              out[ya:yb, xb:]], 1), out[yb:]], 0)
        T = torch.cat([T[:ya], torch.cat([T[ya:yb, :xa], torch.where(blend, Tn, Tw), T[ya:yb, xb:]], 1), T[yb:]], 0)
        done[ya:yb, xa:xb] |= stop
        last[ya:yb, xa:xb] = torch.where(blend, torch.full_like(last[ya:yb, xa:xb], rank), last[ya:yb, xa:xb])
    alpha_img = 1.0 - T
    if background is not None:
        out = out + T[..., None] * background.to(F64)
    return out, alpha_img, last


def rasterization(means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor, viewmats: Tensor,
                  Ks: Tensor, width: int, height: int, backgrounds: Optional[Tensor] = None, render_mode: str = "RGB",
                  near_plane: float = 0.01, far_plane: float = 1e10, eps2d: float = 0.3, **_ignored):
    """The subset of gsplat.rendering.rasterization() the reference uses, C = 1 camera, float64:
    -> (image [1,H,W,D(+1)], alpha [1,H,W,1], meta with the projection outputs)."""
    assert viewmats.shape[0] == 1, "one camera"
    radii, m2d, depths, conics = project(means, quats, scales, viewmats[0], Ks[0], width, height, eps2d, near_plane,
                                         far_plane)
    cols = colors.to(F64)
    bg = backgrounds[0].to(F64) if backgrounds is not None else None
    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat([cols, depths[:, None]], -1)
        if bg is not None:
            bg = torch.cat([bg, torch.zeros(1, dtype=F64)])
    img, alpha, last = composite(m2d, conics, cols, opacities, depths, radii, width, height, bg)
    if render_mode == "RGB+ED":
        img = torch.cat([img[..., :-1], img[..., -1:] / alpha.clamp(min=1e-10)[..., None]], -1)
    return img[None], alpha[None, ..., None], {"radii": radii[None], "means2d": m2d[None], "depths": depths[None],
                                               "conics": conics[None], "last": last[None]}
