"""CPU restatement of main_utils.get_normals (/root/reference/main_utils.py:95-141).  TEST INFRASTRUCTURE ONLY.
Pinned by tests/golden/normals.npz (output and depth gradient of the reference's own function)."""
from __future__ import annotations

import torch


def get_normals(z: torch.Tensor, fx, fy, cx, cy, skew=0.0, offset=0.5) -> torch.Tensor:
    """z [1,H,W] -> [1,3,H,W]."""
    H, W = z.shape[-2:]
    jj, ii = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="xy")
    y = (ii + offset - cy) / fy
    x = (jj + offset - cx - y * skew) / fx
    viewdirs = torch.stack([x, y, torch.ones_like(x)], dim=-1)  # :97-102
    coords = (viewdirs[None] * z[..., None]).squeeze(0)          # :104, :129
    hd, wd, _ = coords.shape
    bottom = coords[2:hd, 1:wd - 1, :]
    top = coords[0:hd - 2, 1:wd - 1, :]
    right = coords[1:hd - 1, 2:wd, :]
    left = coords[1:hd - 1, 0:wd - 2, :]
    n = torch.cross(right - left, top - bottom, dim=-1)          # :135-137
    n = torch.nn.functional.normalize(n, p=2, dim=-1)
    return torch.nn.functional.pad(n.permute(2, 0, 1), (1, 1, 1, 1), mode="constant")[None]  # :139-140
