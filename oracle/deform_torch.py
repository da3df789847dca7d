"""CPU ORACLE (test infrastructure, NOT product code) -- plain-PyTorch restatement of the reference's deformation
network forward:

    deform_network.forward -> forward_dynamic2      /root/reference/scene/deformation.py:252-253, 285-290
    Deformation.forward_dynamic2 / query_time       /root/reference/scene/deformation.py:78-88, 158-199
    HexPlaneField.forward / interpolate_ms_features /root/reference/scene/hexplane.py:19-21, 75-108, 165-187
    quat2mat                                        /root/reference/scene/deformation.py:417-438
    batch_quaternion_multiply                       /root/reference/utils/graphics_utils.py:117-140

PINNED by tests/golden/deform.npz, produced by running the reference's own deform_network in this container
(tests/golden/make_golden.py gen_deform).  Weights are passed as a flat dict (see mobgs_amd.deformation for the
state_dict key mapping).  Only tests/ import this.
"""
from __future__ import annotations

import itertools
import math
from typing import Dict, List

import torch
import torch.nn.functional as F

COMBS = list(itertools.combinations(range(4), 2))  # (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)


def hexplane_features(pts: torch.Tensor, t: torch.Tensor, aabb: torch.Tensor, planes: List[List[torch.Tensor]]):
    """pts [N,3], t [N,1], aabb [2,3] (row 0 = xyz_max, row 1 = xyz_min), planes[level][6] each [1,C,Rb,Ra]."""
    p = torch.clamp((pts - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0, -1.0, 1.0)
    q = torch.cat((p, t), dim=-1)
    feats = []
    for level in planes:
        prod = 1.0
        for ci, (a, b) in enumerate(COMBS):
            grid = level[ci]
            coords = q[:, [a, b]].reshape(1, 1, -1, 2)
            s = F.grid_sample(grid, coords, align_corners=True, mode="bilinear", padding_mode="border")
            prod = prod * s.reshape(grid.shape[1], -1).t()
        feats.append(prod)
    return torch.cat(feats, dim=-1)


def quat2mat5(q4: torch.Tensor) -> torch.Tensor:
    """The reference's quat2mat applied to a 4-vector: prepend 1, divide by the 5-norm, use the first four."""
    nq = torch.cat([q4[:, :1].detach() * 0 + 1, q4], dim=1)
    nq = nq / nq.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = nq[:, 0], nq[:, 1], nq[:, 2], nq[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz, 2 * wz + 2 * xy, w2 - x2 + y2 - z2,
                        2 * yz - 2 * wx, 2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)


def quat_mul_normalized(q1, q2):
    w = q1[:, 0] * q2[:, 0] - q1[:, 1] * q2[:, 1] - q1[:, 2] * q2[:, 2] - q1[:, 3] * q2[:, 3]
    x = q1[:, 0] * q2[:, 1] + q1[:, 1] * q2[:, 0] + q1[:, 2] * q2[:, 3] - q1[:, 3] * q2[:, 2]
    y = q1[:, 0] * q2[:, 2] - q1[:, 1] * q2[:, 3] + q1[:, 2] * q2[:, 0] + q1[:, 3] * q2[:, 1]
    z = q1[:, 0] * q2[:, 3] + q1[:, 1] * q2[:, 2] - q1[:, 2] * q2[:, 1] + q1[:, 3] * q2[:, 0]
    q3 = torch.stack((w, x, y, z), dim=1)
    return q3 / torch.norm(q3, dim=1, keepdim=True)


def mlp_heads(feat, W: Dict[str, torch.Tensor]):
    hidden = F.linear(feat, W["w0"], W["b0"])

    def head(name):
        h = F.linear(F.relu(hidden), W[name + "_w1"], W[name + "_b1"])
        return F.linear(F.relu(h), W[name + "_w2"], W[name + "_b2"])

    return head("pos"), head("scl"), head("rot")


def deform_forward(point, scales, rotations, times_sel, aabb, planes, W):
    feat = hexplane_features(point, times_sel, aabb, planes)
    dx, ds, dr = mlp_heads(feat, W)
    pts = point + dx[:, 0:3]
    pts = quat2mat5(dx[:, 3:]).bmm(pts.view(-1, 3, 1)).view(-1, 3)
    ds = torch.clamp(ds, -math.log(100), math.log(100))
    new_scales = scales + ds
    rot = quat_mul_normalized(rotations + dr, dx[:, 3:])
    return pts, new_scales, rot
