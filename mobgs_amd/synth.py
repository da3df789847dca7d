"""Synthetic scenes with the frozen distributions of SURVEY.md section 8(d) (no dataset ships with the repo).

All draws happen on CPU from a seeded torch.Generator in float32 and are then moved to the target device,
so the CPU oracle and the HIP path see bit-identical inputs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict

import torch


@dataclass
class SynthCamera:
    width: int = 1352
    height: int = 1014
    focal: float = 1170.0
    time: float = 11.0 / 23.0
    max_time: int = 23

    @property
    def K(self) -> torch.Tensor:
        return torch.tensor([[self.focal, 0.0, self.width / 2.0], [0.0, self.focal, self.height / 2.0],
                             [0.0, 0.0, 1.0]], dtype=torch.float32)

    def scaled(self, width: int, height: int) -> "SynthCamera":
        return SynthCamera(width, height, self.focal * width / self.width, self.time, self.max_time)


def gaussian_cloud(n: int, cam: SynthCamera, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Raw (pre-activation) parameters of n splats spread through the camera frustum."""
    g = torch.Generator().manual_seed(seed)
    z = 1.0 + 5.0 * torch.rand(n, generator=g)
    u = 2.0 * torch.rand(n, generator=g) - 1.0
    v = 2.0 * torch.rand(n, generator=g) - 1.0
    x = u * z * (cam.width / (2.0 * cam.focal)) * 1.05
    y = v * z * (cam.height / (2.0 * cam.focal)) * 1.05
    xyz = torch.stack([x, y, z], dim=-1)
    # ~3 px projected sigma (0.0026 * z at the full-size focal 1170), log-normal spread, heavy tail
    scaling = torch.log(0.0026 * (1170.0 / cam.focal) * z)[:, None] + 0.7 * torch.randn(n, 3, generator=g)
    rotation = torch.randn(n, 4, generator=g)
    opacity = 1.5 * torch.randn(n, 1, generator=g)
    a = (2.0 * torch.rand(n, 3, generator=g) - 1.0) * 1.77
    features_dc = torch.cat([a, a + 0.1 * torch.randn(n, 3, generator=g)], dim=-1)
    features_t = 0.1 * torch.randn(n, 3, generator=g)
    return {"xyz": xyz, "scaling": scaling, "rotation": rotation, "opacity": opacity, "features_dc": features_dc,
            "features_t": features_t}


def dynamic_extras(xyz: torch.Tensor, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Extra parameters of the dynamic splats (Hermite control points, angular velocity, time centre)."""
    g = torch.Generator().manual_seed(seed + 7919)
    n = xyz.shape[0]
    omega = 0.05 * torch.randn(n, 4, generator=g)
    trbf_center = torch.rand(n, 1, generator=g)
    walk = torch.cumsum(0.02 * torch.randn(n, 12, 3, generator=g), dim=1)
    control_xyz = 100.0 * (xyz[:, None, :] + walk)
    current_control_num = torch.randint(4, 13, (n, 1), generator=g, dtype=torch.int64)
    return {"omega": omega, "trbf_center": trbf_center, "control_xyz": control_xyz,
            "current_control_num": current_control_num}


def splat_inputs(n: int, cam: SynthCamera, seed: int = 0, channels: int = 9) -> Dict[str, torch.Tensor]:
    """Activated operator-level inputs (what rasterization() receives)."""
    p = gaussian_cloud(n, cam, seed)
    cols = torch.cat([p["features_dc"], p["features_t"]], dim=-1)
    if channels != 9:
        g = torch.Generator().manual_seed(seed + 31)
        cols = torch.randn(n, channels, generator=g)
    return {"means": p["xyz"], "quats": p["rotation"], "scales": torch.exp(p["scaling"]),
            "opacities": torch.sigmoid(p["opacity"]).squeeze(-1), "colors": cols,
            "viewmats": torch.eye(4)[None], "Ks": cam.K[None]}
