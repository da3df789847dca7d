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

Compute: HexPlane gather/product in csrc/deform.hip (32 lanes = 32 channels of a tap); the MLP + update rules in
an MFMA kernel (v_mfma_f32_32x32x2_f32, exact fp32).  Backward: the HexPlane part is a HIP scatter kernel (plane
gradients, point/time gradients); the MLP + update rules back-propagate in csrc/deform_bwd.hip (recomputed hidden
activations, data and weight gradients on fp32 MFMA, per-workgroup partial sums reduced in fixed order).
The plane parameters keep the reference's shape [1,32,rb,ra] but live in torch.channels_last memory format, i.e.
physically [rb][ra][32]: one bilinear tap is one 128-byte row and no per-call re-layout is needed (state_dict /
load_state_dict / optimisers are layout-agnostic).  Only the configuration the reference trains with is supported
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


_view_memo = {}


def _plane_view(p: torch.Tensor) -> torch.Tensor:
    """Physical [rb, ra, 32] view of a [1,32,rb,ra] plane.  For a channels_last tensor it aliases the parameter's
    storage, so the view is remembered per (storage, shape) instead of being rebuilt (3 view ops x 18 planes per call)."""
    if p.dtype == torch.float32 and p.is_contiguous(memory_format=torch.channels_last):
        key = (p.data_ptr(), tuple(p.shape), str(p.device))
        v = _view_memo.get(key)
        if v is None or v[0]() is not p:
            import weakref
            view = p.detach()[0].permute(1, 2, 0)
            if len(_view_memo) > 256:
                _view_memo.clear()
            _view_memo[key] = v = (weakref.ref(p), view)
        return v[1]
    return f32c(p.detach()[0].permute(1, 2, 0))


def _zero_like_planes(planes_cl):
    """Zero gradient planes as views of ONE buffer (one fill launch instead of 18)."""
    sizes = [p.numel() for p in planes_cl]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=planes_cl[0].device)
    out, o = [], 0
    for p, n in zip(planes_cl, sizes):
        out.append(flat[o:o + n].view(p.shape))
        o += n
    return out


class _HexPlane(torch.autograd.Function):
    """pts [N,3], times [N,1], aabb [2,3], 18 planes [1,32,rb,ra] -> features [N,96]."""

    @staticmethod
    def forward(ctx, pts, times, aabb, *planes):
        lib = _lib.load()
        pts, times, aabb = f32c(pts), f32c(times), f32c(aabb)
        if any(p.shape[1] != 32 for p in planes):
            raise NotImplementedError("the HexPlane kernel is built for output_coordinate_dim = 32")
        # [rb, ra, 32] views of channels_last parameters (no copy); any other layout is copied here
        planes_cl = [_plane_view(p) for p in planes]
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
        gplanes = _zero_like_planes(planes_cl)  # one flat zero-filled buffer, 18 views
        v_pts = torch.zeros_like(pts)
        v_times = torch.empty_like(times)
        ptrs, ra, rb = _plane_args(planes_cl)
        gptrs = (ctypes.c_void_p * 18)(*[g.data_ptr() for g in gplanes])
        scratch = torch.empty(lib.mobgs_hexplane_bwd_scratch_bytes(N, ra, rb), dtype=torch.uint8, device=pts.device)
        check(lib.mobgs_hexplane_bwd(N, ptr(pts), ptr(times), ptr(aabb), ptrs, ra, rb, ptr(v_feat), gptrs,
                                     ptr(v_pts), ptr(v_times), ptr(scratch), stream()), "mobgs_hexplane_bwd")
        g_std = [g.permute(2, 0, 1).unsqueeze(0) for g in gplanes]  # back to [1,32,rb,ra]
        return (v_pts, v_times, None, *g_std)


_W_KEYS = ("w0", "b0", "pos_w1", "pos_b1", "pos_w2", "pos_b2", "scl_w1", "scl_b1", "scl_w2", "scl_b2", "rot_w1",
           "rot_b1", "rot_w2", "rot_b2")
_HEADS = ("pos", "scl", "rot")
_NOUT = (7, 3, 4)
# offsets inside the gradient block mobgs_deform_mlp_bwd writes (include/mobgs_hip.h)
_OFF_W0, _OFF_B0, _OFF_W1, _OFF_B1, _OFF_W2, _OFF_B2 = 0, 12288, 12416, 61568, 61952, 66048


def _pack_weights(W, dev):
    """Layouts the kernels read: K-major W0t / W1t / W2t (forward, recompute) and the original (out, in) layouts
    W0 / W1 / W2pad (data gradients)."""
    w0 = f32c(W["w0"].detach())
    w1 = torch.stack([W[h + "_w1"].detach() for h in _HEADS]).contiguous().float()  # [3,128(out),128(in)]
    W2t = torch.zeros(3, 128, 32, dtype=torch.float32, device=dev)
    W2pad = torch.zeros(3, 8, 128, dtype=torch.float32, device=dev)
    b2 = torch.zeros(3, 32, dtype=torch.float32, device=dev)
    for i, h in enumerate(_HEADS):
        w2 = W[h + "_w2"].detach().float()
        W2t[i, :, :_NOUT[i]] = w2.t()
        W2pad[i, :_NOUT[i]] = w2
        b2[i, :_NOUT[i]] = W[h + "_b2"].detach()
    return {"W0t": w0.t().contiguous(), "b0": f32c(W["b0"].detach()), "W1t": w1.transpose(1, 2).contiguous(),
            "b1": torch.stack([W[h + "_b1"].detach() for h in _HEADS]).contiguous().float(), "W2t": W2t, "b2": b2,
            "W0": w0, "W1": w1, "W2pad": W2pad}


_pack_memo = {}


def invalidate_packed_weights() -> None:
    """Forget the re-laid-out MLP weights.  Needed only after edits autograd's version counter does not see -- writes
    through `p.data` (`p.data.copy_()`, manual EMA / weight surgery, old-style optimizers): in-place ops on the
    parameter itself (every torch.optim step, load_state_dict, `with torch.no_grad(): p.mul_()`) bump `_version`
    and re-pack automatically, and a re-allocated `.data` changes the storage pointer that is part of the key."""
    _pack_memo.clear()


def _packed(weights, W, dev):
    """_pack_weights, rebuilt only when a weight tensor was replaced or modified in place (optimizer step):
    ~25 small launches saved per call.  Keyed on identity, storage pointer and autograd version of every weight; see
    invalidate_packed_weights() for the one kind of edit that escapes all three."""
    key = tuple(id(w) for w in weights)
    ver = tuple((w._version, w.data_ptr()) for w in weights)
    hit = _pack_memo.get(key)
    if hit is not None and hit[0] == ver and all(a() is b for a, b in zip(hit[1], weights)):
        return hit[2]
    import weakref
    pk = _pack_weights(W, dev)
    if len(_pack_memo) > 16:
        _pack_memo.clear()
    _pack_memo[key] = (ver, [weakref.ref(w) for w in weights], pk)
    return pk


class _MlpUpdate(torch.autograd.Function):
    """feat [N,96] + (pts, scales, rots) -> (pts', scales', rots'): fp32-MFMA forward and backward."""

    @staticmethod
    def forward(ctx, feat, pts, scales, rots, *weights):
        lib = _lib.load()
        W = dict(zip(_W_KEYS, weights))
        feat, pts, scales, rots = map(f32c, (feat, pts, scales, rots))
        dev = feat.device
        N = feat.shape[0]
        pk = _packed(weights, W, dev)
        need_bwd = any(ctx.needs_input_grad)
        out_pts = torch.empty(N, 3, dtype=torch.float32, device=dev)
        out_scales = torch.empty(N, 3, dtype=torch.float32, device=dev)
        out_rots = torch.empty(N, 4, dtype=torch.float32, device=dev)
        o_raw = torch.empty(N, 16, dtype=torch.float32, device=dev) if need_bwd else None
        check(lib.mobgs_deform_mlp_fwd(N, ptr(feat), ptr(pts), ptr(scales), ptr(rots), ptr(pk["W0t"]), ptr(pk["b0"]),
                                       ptr(pk["W1t"]), ptr(pk["b1"]), ptr(pk["W2t"]), ptr(pk["b2"]), ptr(out_pts),
                                       ptr(out_scales), ptr(out_rots), ptr(o_raw), stream()), "mobgs_deform_mlp_fwd")
        if need_bwd:
            ctx.save_for_backward(feat, pts, rots, o_raw, pk["W0t"], pk["b0"], pk["W1t"], pk["b1"], pk["W0"], pk["W1"],
                                  pk["W2pad"])
        return out_pts, out_scales, out_rots

    @staticmethod
    def backward(ctx, v_pts, v_scales, v_rots):
        lib = _lib.load()
        feat, pts, rots, o_raw, W0t, b0, W1t, b1, W0, W1, W2pad = ctx.saved_tensors
        N, dev = feat.shape[0], feat.device
        c = [f32c(v) if v is not None else None for v in (v_pts, v_scales, v_rots)]

        def E(*shape):
            return torch.empty(*shape, dtype=torch.float32, device=dev)

        g_feat, g_pts, g_rots, v_o = E(N, 96), E(N, 3), E(N, 4), E(N, 16)
        nfl = int(lib.mobgs_deform_mlp_grad_floats())
        partials = E(max(1, lib.mobgs_deform_mlp_bwd_blocks(N)), nfl)
        g = E(nfl)
        check(lib.mobgs_deform_mlp_bwd(N, ptr(feat), ptr(pts), ptr(rots), ptr(o_raw), ptr(W0t), ptr(b0), ptr(W1t),
                                       ptr(b1), ptr(W0), ptr(W1), ptr(W2pad), ptr(c[0]), ptr(c[1]), ptr(c[2]),
                                       ptr(g_feat), ptr(g_pts), ptr(g_rots), ptr(v_o), ptr(partials), ptr(g),
                                       stream()), "mobgs_deform_mlp_bwd")
        gw = {"w0": g[_OFF_W0:_OFF_B0].view(128, 96), "b0": g[_OFF_B0:_OFF_W1]}
        gW2 = g[_OFF_W2:_OFF_B2].view(32, 128)
        gb2 = g[_OFF_B2:_OFF_B2 + 32]
        for i, h in enumerate(_HEADS):
            gw[h + "_w1"] = g[_OFF_W1 + i * 16384:_OFF_W1 + (i + 1) * 16384].view(128, 128)
            gw[h + "_b1"] = g[_OFF_B1 + i * 128:_OFF_B1 + (i + 1) * 128]
            gw[h + "_w2"] = gW2[8 * i:8 * i + _NOUT[i]]
            gw[h + "_b2"] = gb2[8 * i:8 * i + _NOUT[i]]
        g_scales = c[1] if c[1] is not None else torch.zeros(N, 3, dtype=torch.float32, device=dev)
        return (g_feat, g_pts, g_scales, g_rots, *[gw[k] for k in _W_KEYS])


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
                p = nn.Parameter(torch.empty([1, planeconfig["output_coordinate_dim"]] + [reso[c] for c in comb[::-1]])
                                 .contiguous(memory_format=torch.channels_last))  # physically [rb][ra][32]
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


class SeesawArgs:
    """The hidden-model hyper-parameters the reference trains the seesaw scene with
    (/root/reference/arguments/__init__.py:77-107 overridden by arguments/stereo/default.py and seesaw.py)."""
    net_width, timebase_pe, defor_depth, posebase_pe, scale_rotation_pe, opacity_pe = 128, 4, 1, 10, 2, 2
    timenet_width, timenet_output, bounds, grid_pe = 64, 32, 1.6, 0
    kplanes_config = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32,
                      "resolution": [64, 64, 64, 12]}
    multires = [1, 2, 4]
    no_dx = no_grid = no_ds = no_dr = empty_voxel = static_mlp = apply_rotation = False
    no_do = no_dshs = True


def kernel_times(net, pts, scales, rots, times, cots, steps, timer):
    """Per-kernel times of one deform_network call (scripts/bench_deform.py): each C-ABI entry point by itself."""
    lib = _lib.load()
    d = net.deformation_net
    planes_cl = [_plane_view(p) for p in d.grid.planes()]
    aabb = f32c(d.grid.aabb.detach())
    N, dev = pts.shape[0], pts.device
    t1 = f32c(times.reshape(-1, 1))
    ptrs, ra, rb = _plane_args(planes_cl)
    feat = torch.empty(N, 96, dtype=torch.float32, device=dev)
    W = dict(zip(_W_KEYS, d.weights()))
    pk = _pack_weights(W, dev)

    def E(*shape):
        return torch.empty(*shape, dtype=torch.float32, device=dev)

    o_pts, o_scl, o_rot, o_raw = E(N, 3), E(N, 3), E(N, 4), E(N, 16)
    out = {}
    out["hexplane_fwd_ms"] = timer(lambda: check(lib.mobgs_hexplane_fwd(
        N, ptr(pts), ptr(t1), ptr(aabb), ptrs, ra, rb, ptr(feat), stream()), "hexplane_fwd"), steps)
    out["mlp_fwd_ms"] = timer(lambda: check(lib.mobgs_deform_mlp_fwd(
        N, ptr(feat), ptr(pts), ptr(scales), ptr(rots), ptr(pk["W0t"]), ptr(pk["b0"]), ptr(pk["W1t"]), ptr(pk["b1"]),
        ptr(pk["W2t"]), ptr(pk["b2"]), ptr(o_pts), ptr(o_scl), ptr(o_rot), ptr(o_raw), stream()), "mlp_fwd"), steps)
    g_feat, g_pts, g_rots, v_o = E(N, 96), E(N, 3), E(N, 4), E(N, 16)
    nfl = int(lib.mobgs_deform_mlp_grad_floats())
    partials, g = E(lib.mobgs_deform_mlp_bwd_blocks(N), nfl), E(nfl)
    out["mlp_bwd_ms"] = timer(lambda: check(lib.mobgs_deform_mlp_bwd(
        N, ptr(feat), ptr(pts), ptr(rots), ptr(o_raw), ptr(pk["W0t"]), ptr(pk["b0"]), ptr(pk["W1t"]), ptr(pk["b1"]),
        ptr(pk["W0"]), ptr(pk["W1"]), ptr(pk["W2pad"]), ptr(cots[0]), ptr(cots[1]), ptr(cots[2]), ptr(g_feat),
        ptr(g_pts), ptr(g_rots), ptr(v_o), ptr(partials), ptr(g), stream()), "mlp_bwd"), steps)
    gplanes = _zero_like_planes(planes_cl)
    gptrs = (ctypes.c_void_p * 18)(*[x.data_ptr() for x in gplanes])
    v_pts, v_times = torch.zeros_like(pts), torch.empty_like(t1)
    scratch = torch.empty(lib.mobgs_hexplane_bwd_scratch_bytes(N, ra, rb), dtype=torch.uint8, device=dev)
    out["hexplane_bwd_ms"] = timer(lambda: check(lib.mobgs_hexplane_bwd(
        N, ptr(pts), ptr(t1), ptr(aabb), ptrs, ra, rb, ptr(g_feat), gptrs, ptr(v_pts), ptr(v_times), ptr(scratch),
        stream()), "hexplane_bwd"), steps)
    return out


# ---- init-time geometry helpers train.py imports from scene.deformation (train.py:101,113) ------------------------
def _lift_pixels(depth: torch.Tensor, K_inv: torch.Tensor) -> torch.Tensor:
    """depth [B,H,W] -> camera-frame points [B,3,H*W]: depth * K^-1 [u, v, 1]^T at INTEGER pixel coordinates
    (/root/reference/scene/deformation.py:484-506)."""
    B, H, W = depth.shape
    v, u = torch.meshgrid(torch.arange(H, device=depth.device, dtype=depth.dtype),
                          torch.arange(W, device=depth.device, dtype=depth.dtype), indexing="ij")
    pix = torch.stack([u, v, torch.ones_like(u)], dim=0).reshape(1, 3, H * W)
    return torch.bmm(K_inv, pix.expand(B, 3, H * W)) * depth.reshape(B, 1, H * W)


def _camera_to_world(w2c: torch.Tensor, cam_pts: torch.Tensor) -> torch.Tensor:
    R, t = w2c[:, :, 0:3], w2c[:, :, 3:4]
    Rt = R.transpose(1, 2)
    return torch.bmm(Rt, cam_pts) - torch.bmm(Rt, t)


def points_from_DRTK(depth, w2c1, intrinsics):
    """World coordinates of every pixel of a depth map (/root/reference/scene/deformation.py:758-782).
    depth [B,1,H,W], w2c1 [B,3,4], intrinsics [B,3,3] -> [B,3,H*W]."""
    return _camera_to_world(w2c1, _lift_pixels(depth[:, 0], torch.inverse(intrinsics)))


def inverse_warp_rt1_rt2(img, depth, w2c1, w2c2, intrinsics, intrinsics_inv, padding_mode="zeros", ret_grid=False):
    """Sample `img` (seen by camera 2) at the re-projection of camera 1's depth map
    (/root/reference/scene/deformation.py:640-699).  img [B,C,H,W], depth [B,1,H,W], w2c1 / w2c2 [B,3,4]."""
    d = depth[:, 0]
    B, H, W = d.shape
    world = _camera_to_world(w2c1, _lift_pixels(d, intrinsics_inv))
    c2 = torch.bmm(w2c2[:, :, 0:3], world) + w2c2[:, :, 3:4]
    z = c2[:, 2:3, :]
    z = torch.where(z.abs() < 1e-6, torch.full_like(z, 1e-6), z)
    p2 = torch.bmm(intrinsics, c2 / z)
    x = 2 * p2[:, 0] / (W - 1) - 1
    y = 2 * p2[:, 1] / (H - 1) - 1
    if padding_mode == "zeros":  # anything off-image samples well outside: no blend of image and padding
        x = torch.where(((x > 1) | (x < -1)).detach(), torch.full_like(x, 2.0), x)
        y = torch.where(((y > 1) | (y < -1)).detach(), torch.full_like(y, 2.0), y)
    grid = torch.stack([x, y], dim=2).view(B, H, W, 2)
    out = F.grid_sample(img, grid, padding_mode=padding_mode, align_corners=True)
    return (out, grid) if ret_grid else out
