"""CPU ORACLE (test infrastructure, NOT product code) -- pure-PyTorch float32 restatement of the
gsplat v1.4.0 operator API that MoBGS calls.

PARITY UNPINNED vs. real gsplat: gsplat==1.4.0 is a third-party pip dependency of the reference
(/root/reference/README.md:26, imported at /root/reference/gaussian_renderer/__init__.py:15) that
is neither vendored in /root/reference nor installed in this image, and the reference has no
tests / golden vectors for it.  This file restates the published v1.4.0 algorithm
(gsplat/rendering.py, gsplat/cuda/csrc/{fully_fused_projection_fwd,isect_tiles,
rasterize_to_pixels_fwd}.cu, utils.cuh) and is anchored on the
reference's call sites:
    rasterization(...)            /root/reference/gaussian_renderer/__init__.py:143,163,201,236,255,274,
                                  379,437,456,473,538
    fully_fused_projection(...)   /root/reference/gaussian_renderer/__init__.py:190,411,422,513,524
Gradients come from torch autograd over this forward restatement, which makes them an
independent check of the hand-derived backward kernels (HIP and oracle/gsplat_cpu.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.999
T_STOP = 1e-4


# --------------------------------------------------------------------------------------
# A.1 projection  (fully_fused_projection_fwd.cu [upstream])
# --------------------------------------------------------------------------------------
def quat_to_rotmat(quats: Tensor) -> Tensor:
    """(w,x,y,z), normalised inside (utils.cuh quat_to_rotmat [upstream])."""
    q = quats * torch.rsqrt((quats * quats).sum(-1, keepdim=True))
    w, x, y, z = q.unbind(-1)
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )
    return R.reshape(quats.shape[:-1] + (3, 3))


def fully_fused_projection(
    means: Tensor,  # [N,3]
    covars: Optional[Tensor],
    quats: Optional[Tensor],  # [N,4]
    scales: Optional[Tensor],  # [N,3]
    viewmats: Tensor,  # [C,4,4]
    Ks: Tensor,  # [C,3,3]
    width: int,
    height: int,
    eps2d: float = 0.3,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    packed: bool = False,
    sparse_grad: bool = False,
    calc_compensations: bool = False,
    camera_model: str = "pinhole",
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Optional[Tensor]]:
    assert covars is None and not packed and camera_model == "pinhole"
    C = viewmats.shape[0]
    R = viewmats[:, :3, :3]  # [C,3,3]
    t = viewmats[:, :3, 3]  # [C,3]
    mean_c = torch.einsum("cij,nj->cni", R, means) + t[:, None, :]  # [C,N,3]
    Rq = quat_to_rotmat(quats)  # [N,3,3]
    M = Rq * scales[:, None, :]
    covar = M @ M.transpose(-1, -2)  # [N,3,3]
    covar_c = torch.einsum("cij,njk,clk->cnil", R, covar, R)  # [C,N,3,3]

    fx, fy = Ks[:, 0, 0][:, None], Ks[:, 1, 1][:, None]
    cx, cy = Ks[:, 0, 2][:, None], Ks[:, 1, 2][:, None]
    x, y, z = mean_c.unbind(-1)
    valid = (z >= near_plane) & (z <= far_plane)
    zs = torch.where(valid, z, torch.ones_like(z))
    tan_fovx = 0.5 * width / fx
    tan_fovy = 0.5 * height / fy
    lim_x_pos = (width - cx) / fx + 0.3 * tan_fovx
    lim_x_neg = cx / fx + 0.3 * tan_fovx
    lim_y_pos = (height - cy) / fy + 0.3 * tan_fovy
    lim_y_neg = cy / fy + 0.3 * tan_fovy
    rz = 1.0 / zs
    rz2 = rz * rz
    tx = zs * torch.minimum(lim_x_pos, torch.maximum(-lim_x_neg, x * rz))
    ty = zs * torch.minimum(lim_y_pos, torch.maximum(-lim_y_neg, y * rz))
    O = torch.zeros_like(z)
    J = torch.stack([fx * rz, O, -fx * tx * rz2, O, fy * rz, -fy * ty * rz2], dim=-1).reshape(
        C, -1, 2, 3
    )
    cov2d = J @ covar_c @ J.transpose(-1, -2)  # [C,N,2,2]
    mean2d = torch.stack([fx * x * rz + cx, fy * y * rz + cy], dim=-1)

    a = cov2d[..., 0, 0] + eps2d
    b = cov2d[..., 0, 1]
    d = cov2d[..., 1, 1] + eps2d
    det = a * d - b * b
    valid = valid & (det > 0)
    dets = torch.where(valid, det, torch.ones_like(det))
    conics = torch.stack([d / dets, -b / dets, a / dets], dim=-1)
    with torch.no_grad():
        bb = 0.5 * (a + d)
        v1 = bb + torch.sqrt(torch.clamp(bb * bb - det, min=0.01))
        radius = torch.ceil(3.0 * torch.sqrt(v1))
        valid = valid & (radius > radius_clip)
        valid = valid & ~(
            (mean2d[..., 0] + radius <= 0)
            | (mean2d[..., 0] - radius >= width)
            | (mean2d[..., 1] + radius <= 0)
            | (mean2d[..., 1] - radius >= height)
        )
        radii = torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32)
    vm = valid[..., None]
    means2d = torch.where(vm, mean2d, torch.zeros_like(mean2d))
    depths = torch.where(valid, z, torch.zeros_like(z))
    conics = torch.where(vm, conics, torch.zeros_like(conics))
    return radii, means2d, depths, conics, None


# --------------------------------------------------------------------------------------
# A.2 tile intersection + ordering (isect_tiles.cu + CUB radix sort [upstream])
# --------------------------------------------------------------------------------------
@torch.no_grad()
def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int,
                tile_height: int):
    """Returns tiles_per_gauss [C,N] i32, isect_ids i64 [I] (sorted), flatten_ids i32 [I] (sorted)."""
    C, N = radii.shape
    ts = float(tile_size)
    r = radii.to(torch.float32) / ts
    tx = means2d[..., 0] / ts
    ty = means2d[..., 1] / ts
    x0 = torch.clamp(torch.floor(tx - r), 0, tile_width).to(torch.int64)
    x1 = torch.clamp(torch.ceil(tx + r), 0, tile_width).to(torch.int64)
    y0 = torch.clamp(torch.floor(ty - r), 0, tile_height).to(torch.int64)
    y1 = torch.clamp(torch.ceil(ty + r), 0, tile_height).to(torch.int64)
    vis = radii > 0
    nx = torch.where(vis, x1 - x0, torch.zeros_like(x0))
    ny = torch.where(vis, y1 - y0, torch.zeros_like(y0))
    tiles_per_gauss = (nx * ny).to(torch.int32)
    cnt = tiles_per_gauss.reshape(-1).to(torch.int64)
    I = int(cnt.sum())
    owner = torch.repeat_interleave(torch.arange(C * N), cnt)  # flat c*N+i per intersection
    start = torch.cumsum(cnt, 0) - cnt
    k = torch.arange(I) - start[owner]
    w = nx.reshape(-1)[owner]
    tyi = y0.reshape(-1)[owner] + k // torch.clamp(w, min=1)
    txi = x0.reshape(-1)[owner] + k % torch.clamp(w, min=1)
    cam = owner // N
    tile_id = tyi * tile_width + txi
    n_tiles = tile_width * tile_height
    tile_bits = max(1, int(math.floor(math.log2(n_tiles))) + 1)
    depth_bits = depths.reshape(-1)[owner].to(torch.float32).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    keys = (cam << (32 + tile_bits)) | (tile_id << 32) | depth_bits
    keys_sorted, perm = torch.sort(keys, stable=True)
    flatten_ids = owner[perm].to(torch.int32)
    return tiles_per_gauss, keys_sorted, flatten_ids


@torch.no_grad()
def isect_offset_encode(isect_ids: Tensor, C: int, tile_width: int, tile_height: int) -> Tensor:
    n_tiles = tile_width * tile_height
    tile_bits = max(1, int(math.floor(math.log2(n_tiles))) + 1)
    cam = isect_ids >> (32 + tile_bits)
    tile = (isect_ids >> 32) & ((1 << tile_bits) - 1)
    flat = cam * n_tiles + tile
    offs = torch.searchsorted(flat.contiguous(), torch.arange(C * n_tiles))
    return offs.to(torch.int32).reshape(C, tile_height, tile_width)


# --------------------------------------------------------------------------------------
# A.3 rasterize_to_pixels forward (autograd supplies A.4)
# --------------------------------------------------------------------------------------
def rasterize_to_pixels(
    means2d: Tensor,  # [C,N,2]
    conics: Tensor,  # [C,N,3]
    colors: Tensor,  # [C,N,D]
    opacities: Tensor,  # [C,N]
    image_width: int,
    image_height: int,
    tile_size: int,
    isect_offsets: Tensor,  # [C,th,tw]
    flatten_ids: Tensor,  # [I]
    backgrounds: Optional[Tensor] = None,  # [C,D]
    return_last_ids: bool = False,
):
    C, N = means2d.shape[:2]
    D = colors.shape[-1]
    th, tw = isect_offsets.shape[1:]
    I = flatten_ids.shape[0]
    offs = torch.cat([isect_offsets.reshape(-1).to(torch.int64), torch.tensor([I])])
    m2 = means2d.reshape(C * N, 2)
    cn = conics.reshape(C * N, 3)
    cl = colors.reshape(C * N, D)
    op = opacities.reshape(C * N)
    fid = flatten_ids.to(torch.int64)
    out_rows = []
    alpha_rows = []
    last_rows = []
    dev = means2d.device
    ly, lx = torch.meshgrid(torch.arange(tile_size), torch.arange(tile_size), indexing="ij")
    for c in range(C):
        img = torch.zeros(th * tile_size, tw * tile_size, D, dtype=colors.dtype, device=dev)
        alp = torch.zeros(th * tile_size, tw * tile_size, dtype=colors.dtype, device=dev)
        last = torch.zeros(th * tile_size, tw * tile_size, dtype=torch.int32, device=dev)
        tiles_c = []
        for tyi in range(th):
            row_c = []
            row_a = []
            for txi in range(tw):
                tflat = (c * th + tyi) * tw + txi
                s, e = int(offs[tflat]), int(offs[tflat + 1])
                px = (txi * tile_size + lx).reshape(-1).to(colors.dtype) + 0.5
                py = (tyi * tile_size + ly).reshape(-1).to(colors.dtype) + 0.5
                if e <= s:
                    pc = torch.zeros(tile_size * tile_size, D, dtype=colors.dtype)
                    T_fin = torch.ones(tile_size * tile_size, dtype=colors.dtype)
                    lastk = torch.zeros(tile_size * tile_size, dtype=torch.int64)
                else:
                    g = fid[s:e]
                    xy = m2[g]
                    co = cn[g]
                    dx = xy[None, :, 0] - px[:, None]
                    dy = xy[None, :, 1] - py[:, None]
                    sigma = 0.5 * (co[None, :, 0] * dx * dx + co[None, :, 2] * dy * dy) + co[None, :, 1] * dx * dy
                    alpha = torch.clamp(op[g][None, :] * torch.exp(-sigma), max=ALPHA_MAX)
                    valid = (sigma >= 0) & (alpha >= ALPHA_MIN)
                    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
                    with torch.no_grad():
                        T_incl = torch.cumprod(1 - a_eff, dim=1)
                        stop = valid & (T_incl <= T_STOP)
                        done = torch.cummax(stop.to(torch.int8), dim=1).values.bool()
                        contrib = valid & ~done
                    a_inc = torch.where(contrib, alpha, torch.zeros_like(alpha))
                    T_incl2 = torch.cumprod(1 - a_inc, dim=1)
                    T_excl = torch.cat([torch.ones_like(T_incl2[:, :1]), T_incl2[:, :-1]], dim=1)
                    wgt = a_inc * T_excl
                    pc = wgt @ cl[g]
                    T_fin = T_incl2[:, -1]
                    with torch.no_grad():
                        idx = torch.arange(s, e)[None, :].expand_as(contrib)
                        lastk = torch.where(contrib, idx, torch.zeros_like(idx)).max(dim=1).values
                if backgrounds is not None:
                    pc = pc + T_fin[:, None] * backgrounds[c][None, :]
                row_c.append(pc.reshape(tile_size, tile_size, D))
                row_a.append((1 - T_fin).reshape(tile_size, tile_size))
                last[tyi * tile_size:(tyi + 1) * tile_size, txi * tile_size:(txi + 1) * tile_size] = lastk.reshape(
                    tile_size, tile_size).to(torch.int32)
            tiles_c.append((torch.cat(row_c, dim=1), torch.cat(row_a, dim=1)))
        img = torch.cat([t[0] for t in tiles_c], dim=0)[:image_height, :image_width]
        alp = torch.cat([t[1] for t in tiles_c], dim=0)[:image_height, :image_width]
        out_rows.append(img)
        alpha_rows.append(alp)
        last_rows.append(last[:image_height, :image_width])
    render_colors = torch.stack(out_rows, 0)
    render_alphas = torch.stack(alpha_rows, 0)[..., None]
    if return_last_ids:
        return render_colors, render_alphas, torch.stack(last_rows, 0)
    return render_colors, render_alphas


# --------------------------------------------------------------------------------------
# rasterization() wrapper (gsplat/rendering.py [upstream]) -- only the options MoBGS uses
# --------------------------------------------------------------------------------------
def rasterization(
    means: Tensor,
    quats: Tensor,
    scales: Tensor,
    opacities: Tensor,
    colors: Tensor,
    viewmats: Tensor,
    Ks: Tensor,
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
    distributed: bool = False,
    camera_model: str = "pinhole",
    covars: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Dict]:
    assert not packed and sh_degree is None and covars is None and not absgrad
    assert rasterize_mode == "classic" and camera_model == "pinhole" and not distributed
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED")
    C = viewmats.shape[0]
    N = means.shape[0]
    radii, means2d, depths, conics, _ = fully_fused_projection(
        means, None, quats, scales, viewmats, Ks, width, height, eps2d=eps2d, near_plane=near_plane,
        far_plane=far_plane, radius_clip=radius_clip)
    opac = opacities[None].expand(C, N)
    if colors.dim() == 2:
        cols = colors[None].expand(C, N, colors.shape[-1])
    else:
        cols = colors
    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat([cols, depths[..., None]], dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros(C, 1, dtype=backgrounds.dtype)], dim=-1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros(C, 1, dtype=backgrounds.dtype)
    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, tile_size, tile_width,
                                                          tile_height)
    isect_offsets = isect_offset_encode(isect_ids, C, tile_width, tile_height)
    render_colors, render_alphas = rasterize_to_pixels(
        means2d, conics, cols, opac, width, height, tile_size, isect_offsets, flatten_ids,
        backgrounds=backgrounds)
    if render_mode in ("ED", "RGB+ED"):
        render_colors = torch.cat(
            [render_colors[..., :-1], render_colors[..., -1:] / render_alphas.clamp(min=1e-10)], dim=-1)
    meta = {
        "camera_ids": None, "gaussian_ids": None, "radii": radii, "means2d": means2d, "depths": depths,
        "conics": conics, "opacities": opac, "tile_width": tile_width, "tile_height": tile_height,
        "tiles_per_gauss": tiles_per_gauss, "isect_ids": isect_ids, "flatten_ids": flatten_ids,
        "isect_offsets": isect_offsets, "width": width, "height": height, "tile_size": tile_size,
        "n_cameras": C,
    }
    return render_colors, render_alphas, meta
