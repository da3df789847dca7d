"""Deformation API (boundary B1): drop-in for the reference's `scene.deformation.deform_network`.

    from mobgs_amd.deformation import deform_network
    net = deform_network(args); net.deformation_net.set_aabb(xyz_max, xyz_min)
    pts, scales, rotations = net(point, scales, rotations, times_sel)

mirrors /root/reference/scene/deformation.py:228-303 (deform_network), :18-199 (Deformation.forward_dynamic2) and
/root/reference/scene/hexplane.py:112-187 (HexPlaneField).  The module tree and parameter names are the reference's
(`deformation_net.grid.grids.<level>.<plane>`, `deformation_net.feature_out.0`, `deformation_net.pos_deform.{1,3}`,
`scales_deform`, `rotations_deform`, `timenet`, the *_poc buffers), so `deformation.pth` checkpoints load unchanged.

SURVEY.md section 0, surprise #1: the reference builds, optimises and checkpoints this network but never calls it
from render(); it is provided because the north star names it (BASELINE config #3).

Compute: HexPlane gather/product in csrc/deform.hip (channels-last planes, 32 lanes = 32 channels of a tap); the
MLP + update rules in an MFMA kernel (v_mfma_f32_32x32x2_f32, exact fp32).  Backward: the HexPlane part is a HIP
scatter kernel (plane gradients, point/time gradients); the MLP + update rules back-propagate through plain
rocBLAS GEMMs (torch.matmul) recomputed from the saved 96-float feature rows -- a hand-written MFMA backward is
the next step.  Only the configuration the reference trains with is supported
(no_grid=False, grid_pe=0, static_mlp=False, empty_voxel=False, defor_depth=1, no_dx/no_ds/no_dr=False,
apply_rotation=False): anything else raises NotImplementedError.
"""
from __future__ import annotations

import ctypes
import itertools
import math
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.init as init

from . import _lib
from ._lib import check, f32c, ptr, stream

COMBS = list(itertools.combinations(range(4), 2))
LOG100 = math.log(100)


def _plane_args(planes_cl: List[torch.Tensor]):
    """HOST arrays for the C ABI: 18 device pointers + widths + heights of channels-last planes [rb, ra, 32]."""
    assert len(planes_cl) == 18
    ptrs = (ctypes.c_void_p * 18)(*[p.data_ptr() for p in planes_cl])
    ra = (ctypes.c_int32 * 18)(*[p.shape[1] for p in planes_cl])
    rb = (ctypes.c_int32 * 18)(*[p.shape[0] for p in planes_cl])
    return ptrs, ra, rb


class _HexPlane(torch.autograd.Function):
    """pts [N,3], times [N,1], aabb [2,3], 18 planes [1,32,rb,ra] -> features [N,96]."""

    @staticmethod
    def forward(ctx, pts, times, aabb, *planes):
        lib = _lib.load()
        pts, times, aabb = f32c(pts), f32c(times), f32c(aabb)
        if any(p.shape[1] != 32 for p in planes):
            raise NotImplementedError("the HexPlane kernel is built for output_coordinate_dim = 32")
        planes_cl = [f32c(p[0].permute(1, 2, 0)) for p in planes]  # [rb, ra, 32]
        N = pts.shape[0]
        feat = torch.empty(N, 96, dtype=torch.float32, device=pts.device)
        ptrs, ra, rb = _plane_args(planes_cl)
        check(lib.mobgs_hexplane_fwd(N, ptr(pts), ptr(times), ptr(aabb), ptrs, ra, rb, ptr(feat), stream()),
              "mobgs_hexplane_fwd")
        ctx.save_for_backward(pts, times, aabb, *planes_cl)
        return feat

    @staticmethod
    def backward(ctx, v_feat):
        lib = _lib.load()
        pts, times, aabb, *planes_cl = ctx.saved_tensors
        N = pts.shape[0]
        v_feat = f32c(v_feat)
        gplanes = [torch.zeros_like(p) for p in planes_cl]
        v_pts = torch.zeros_like(pts)
        v_times = torch.empty_like(times)
        ptrs, ra, rb = _plane_args(planes_cl)
        gptrs = (ctypes.c_void_p * 18)(*[g.data_ptr() for g in gplanes])
        check(lib.mobgs_hexplane_bwd(N, ptr(pts), ptr(times), ptr(aabb), ptrs, ra, rb, ptr(v_feat), gptrs,
                                     ptr(v_pts), ptr(v_times), stream()), "mobgs_hexplane_bwd")
        g_std = [g.permute(2, 0, 1).unsqueeze(0) for g in gplanes]  # back to [1,32,rb,ra]
        return (v_pts, v_times, None, *g_std)


def _mlp_update_torch(feat, pts, scales, rots, W):
    """MLP heads + update rules with library GEMMs (used for the backward pass only)."""
    hidden = F.linear(feat, W["w0"], W["b0"])

    def head(n):
        return F.linear(F.relu(F.linear(F.relu(hidden), W[n + "_w1"], W[n + "_b1"])), W[n + "_w2"], W[n + "_b2"])

    dx, ds, dr = head("pos"), head("scl"), head("rot")
    p = pts + dx[:, 0:3]
    nq = torch.cat([torch.ones_like(dx[:, :1]), dx[:, 3:]], dim=1)
    nq = nq / nq.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = nq[:, 0], nq[:, 1], nq[:, 2], nq[:, 3]
    R = torch.stack([w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
                     2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
                     2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z], dim=1).view(-1, 3, 3)
    out_pts = R.bmm(p.unsqueeze(-1)).squeeze(-1)
    out_scales = scales + torch.clamp(ds, -LOG100, LOG100)
    q1, q2 = rots + dr, dx[:, 3:]
    q3 = torch.stack((q1[:, 0] * q2[:, 0] - q1[:, 1] * q2[:, 1] - q1[:, 2] * q2[:, 2] - q1[:, 3] * q2[:, 3],
                      q1[:, 0] * q2[:, 1] + q1[:, 1] * q2[:, 0] + q1[:, 2] * q2[:, 3] - q1[:, 3] * q2[:, 2],
                      q1[:, 0] * q2[:, 2] - q1[:, 1] * q2[:, 3] + q1[:, 2] * q2[:, 0] + q1[:, 3] * q2[:, 1],
                      q1[:, 0] * q2[:, 3] + q1[:, 1] * q2[:, 2] - q1[:, 2] * q2[:, 1] + q1[:, 3] * q2[:, 0]), dim=1)
    return out_pts, out_scales, q3 / torch.norm(q3, dim=1, keepdim=True)


_W_KEYS = ("w0", "b0", "pos_w1", "pos_b1", "pos_w2", "pos_b2", "scl_w1", "scl_b1", "scl_w2", "scl_b2", "rot_w1",
           "rot_b1", "rot_w2", "rot_b2")


class _MlpUpdate(torch.autograd.Function):
    """feat [N,96] + (pts, scales, rots) -> (pts', scales', rots'): MFMA forward, library-GEMM backward."""

    @staticmethod
    def forward(ctx, feat, pts, scales, rots, *weights):
        lib = _lib.load()
        W = dict(zip(_W_KEYS, weights))
        feat, pts, scales, rots = map(f32c, (feat, pts, scales, rots))
        dev = feat.device
        N = feat.shape[0]
        W0t = f32c(W["w0"].t())  # [96,128] K-major
        heads = ("pos", "scl", "rot")
        W1t = torch.stack([W[h + "_w1"].t() for h in heads]).contiguous()  # [3,128,128]
        b1 = torch.stack([W[h + "_b1"] for h in heads]).contiguous()
        W2t = torch.zeros(3, 128, 32, dtype=torch.float32, device=dev)
        b2 = torch.zeros(3, 32, dtype=torch.float32, device=dev)
        for i, h in enumerate(heads):
            n = W[h + "_w2"].shape[0]
            W2t[i, :, :n] = W[h + "_w2"].t()
            b2[i, :n] = W[h + "_b2"]
        out_pts = torch.empty(N, 3, dtype=torch.float32, device=dev)
        out_scales = torch.empty(N, 3, dtype=torch.float32, device=dev)
        out_rots = torch.empty(N, 4, dtype=torch.float32, device=dev)
        check(lib.mobgs_deform_mlp_fwd(N, ptr(feat), ptr(pts), ptr(scales), ptr(rots), ptr(W0t), ptr(f32c(W["b0"])),
                                       ptr(W1t), ptr(b1), ptr(W2t), ptr(b2), ptr(out_pts), ptr(out_scales),
                                       ptr(out_rots), stream()), "mobgs_deform_mlp_fwd")
        ctx.save_for_backward(feat, pts, scales, rots, *weights)
        return out_pts, out_scales, out_rots

    @staticmethod
    def backward(ctx, v_pts, v_scales, v_rots):
        feat, pts, scales, rots, *weights = ctx.saved_tensors
        with torch.enable_grad():
            leaves = [t.detach().requires_grad_(True) for t in (feat, pts, scales, rots, *weights)]
            outs = _mlp_update_torch(*leaves[:4], dict(zip(_W_KEYS, leaves[4:])))
            cots = [v if v is not None else torch.zeros_like(o) for v, o in zip((v_pts, v_scales, v_rots), outs)]
            grads = torch.autograd.grad(outs, leaves, cots, allow_unused=True)
        return tuple(grads)


class HexPlaneField(nn.Module):
    """/root/reference/scene/hexplane.py:112-187 (same parameters, same aabb convention)."""

    def __init__(self, bounds, planeconfig, multires):
        super().__init__()
        self.aabb = nn.Parameter(torch.tensor([[bounds] * 3, [-bounds] * 3], dtype=torch.float32), requires_grad=False)
        self.grid_config = [planeconfig]
        self.multiscale_res_multipliers = multires
        self.concat_features = True
        self.grids = nn.ModuleList()
        self.feat_dim = 0
        for res in multires:
            reso = [r * res for r in planeconfig["resolution"][:3]] + list(planeconfig["resolution"][3:])
            gp = nn.ParameterList()
            for comb in COMBS:
                p = nn.Parameter(torch.empty([1, planeconfig["output_coordinate_dim"]] + [reso[c] for c in comb[::-1]]))
                if 3 in comb:
                    nn.init.ones_(p)  # time planes start at 1
                else:
                    nn.init.uniform_(p, a=0.1, b=0.5)
                gp.append(p)
            self.feat_dim += planeconfig["output_coordinate_dim"]
            self.grids.append(gp)

    @property
    def get_aabb(self):
        return self.aabb[0], self.aabb[1]

    def set_aabb(self, xyz_max, xyz_min, ref_type=None):
        aabb = torch.tensor([xyz_max, xyz_min], dtype=torch.float32, device=self.aabb.device)
        self.aabb = nn.Parameter(aabb, requires_grad=False)

    def planes(self):
        return [p for level in self.grids for p in level]

    def forward(self, pts, timestamps=None):
        if len(self.grids) != 3:
            raise NotImplementedError("the HexPlane kernel is built for 3 resolution levels (multires of length 3)")
        return _HexPlane.apply(pts.reshape(-1, 3), timestamps.reshape(-1, 1), self.aabb, *self.planes())


class Deformation(nn.Module):
    """/root/reference/scene/deformation.py:18-199 for the configuration the reference trains with."""

    def __init__(self, D=8, W=256, input_ch=27, input_ch_time=9, grid_pe=0, skips=(), args=None):
        super().__init__()
        unsupported = (args.no_grid or args.empty_voxel or args.static_mlp or grid_pe != 0 or D != 1 or args.no_dx
                       or args.no_ds or args.no_dr or args.apply_rotation or W != 128)
        if unsupported:
            raise NotImplementedError("mobgs_amd.deformation supports the reference's training configuration only "
                                      "(net_width=128, defor_depth=1, grid on, dx/ds/dr heads, apply_rotation=False)")
        self.D, self.W, self.args = D, W, args
        self.grid = HexPlaneField(args.bounds, args.kplanes_config, args.multires)
        if self.grid.feat_dim != 96:
            raise NotImplementedError("HexPlane feature width must be 96 (3 levels x 32 channels)")
        self.ratio = 0
        self.feature_out = nn.Sequential(nn.Linear(self.grid.feat_dim, W))
        self.pos_deform = nn.Sequential(nn.ReLU(), nn.Linear(W, W), nn.ReLU(), nn.Linear(W, 7))
        self.scales_deform = nn.Sequential(nn.ReLU(), nn.Linear(W, W), nn.ReLU(), nn.Linear(W, 3))
        self.rotations_deform = nn.Sequential(nn.ReLU(), nn.Linear(W, W), nn.ReLU(), nn.Linear(W, 4))

    @property
    def get_aabb(self):
        return self.grid.get_aabb

    def set_aabb(self, xyz_max, xyz_min, ref_type=None):
        self.grid.set_aabb(xyz_max, xyz_min, ref_type)

    def weights(self):
        out = [self.feature_out[0].weight, self.feature_out[0].bias]
        for seq in (self.pos_deform, self.scales_deform, self.rotations_deform):
            out += [seq[1].weight, seq[1].bias, seq[3].weight, seq[3].bias]
        return out

    def forward_dynamic2(self, rays_pts_emb, scales_emb, rotations_emb, time_emb):
        pts, scales, rots = rays_pts_emb[:, :3], scales_emb[:, :3], rotations_emb[:, :4]
        feat = self.grid(pts, time_emb[:, :1])
        return _MlpUpdate.apply(feat, pts, scales, rots, *self.weights())

    def get_mlp_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" not in n]

    def get_grid_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" in n]


def initialize_weights(m):
    if isinstance(m, nn.Linear):
        init.xavier_uniform_(m.weight, gain=1)  # biases keep PyTorch's default (scene/deformation.py:786-792)


class deform_network(nn.Module):
    def __init__(self, args):
        super().__init__()
        times_ch = 2 * args.timebase_pe + 1
        self.timenet = nn.Sequential(nn.Linear(times_ch, args.timenet_width), nn.ReLU(),
                                     nn.Linear(args.timenet_width, args.timenet_output))
        self.deformation_net = Deformation(W=args.net_width, D=args.defor_depth,
                                           input_ch=3 + 3 * args.posebase_pe * 2, grid_pe=args.grid_pe,
                                           input_ch_time=args.timenet_output, args=args)
        self.register_buffer("time_poc", torch.FloatTensor([2 ** i for i in range(args.timebase_pe)]))
        self.register_buffer("pos_poc", torch.FloatTensor([2 ** i for i in range(args.posebase_pe)]))
        self.register_buffer("rotation_scaling_poc", torch.FloatTensor([2 ** i for i in range(args.scale_rotation_pe)]))
        self.register_buffer("opacity_poc", torch.FloatTensor([2 ** i for i in range(args.opacity_pe)]))
        self.apply(initialize_weights)

    def forward(self, point, scales, rotations, times_sel):
        return self.forward_dynamic2(point, scales, rotations, times_sel)

    @property
    def get_aabb(self):
        return self.deformation_net.get_aabb

    @property
    def get_empty_ratio(self):
        return self.deformation_net.ratio

    def forward_dynamic2(self, point, scales=None, rotations=None, times_sel=None):
        # the reference builds sin/cos embeddings here (poc_fre, :286-288) and then only consumes their raw
        # leading columns (:172,:187,:196): dead compute, skipped
        return self.deformation_net.forward_dynamic2(point, scales, rotations, times_sel)

    def get_mlp_parameters(self):
        return self.deformation_net.get_mlp_parameters() + list(self.timenet.parameters())

    def get_grid_parameters(self):
        return self.deformation_net.get_grid_parameters()
