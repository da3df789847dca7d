"""Colour decoder module with the reference's parameter layout.

`Sandwich` keeps the state_dict keys (mlp1.weight [6,12,1,1], mlp2.weight [3,6,1,1], no bias) of
/root/reference/helper_model.py:7-28 so `point_cloud.pt` decoder checkpoints load unchanged; its forward runs the
fused gfx950 decoder kernel (mobgs_amd.ops.decode).
"""
from __future__ import annotations

import torch
import torch.nn as nn


class Sandwich(nn.Module):
    def __init__(self, dim: int = 9, outdim: int = 3, bias: bool = False):
        super().__init__()
        if bias:
            raise NotImplementedError("the reference never enables the decoder bias")
        self.mlp1 = nn.Conv2d(12, 6, kernel_size=1, bias=False)
        self.mlp2 = nn.Conv2d(6, 3, kernel_size=1, bias=False)

    def forward(self, input: torch.Tensor, rays: torch.Tensor, time=None) -> torch.Tensor:  # noqa: A002
        """input [1,9,H,W] (albedo|spec|time feature), rays [1,6,H,W] -> rgb [1,3,H,W]"""
        from .ops import decode_nchw
        return decode_nchw(input, rays, self.mlp1.weight, self.mlp2.weight)


def getcolormodel(rgbfuntion: str):
    if rgbfuntion == "sandwich":
        return Sandwich(9, 3)
    raise NotImplementedError(f"colour model {rgbfuntion!r}: only 'sandwich' is used by the reference")
