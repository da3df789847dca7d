"""autograd wrappers of the fused per-splat and per-pixel kernels (prep.hip, decoder.hip)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import _lib
from . import _fast
from ._lib import stream_int  # noqa: E402
from ._lib import attr_c, check, f32c, ptr, stream


_LEAF_NAMES = ("s_xyz", "s_scaling", "s_rotation", "s_opacity", "s_fdc", "s_ft", "d_control", "d_scaling",
               "d_rotation", "d_omega", "d_opacity", "d_fdc", "d_ft")
_active_sink = None


class LeafGradSink:
    """Optional: one set of gradient buffers for all renders of one backward pass.

        with LeafGradSink(stat_pc, dyn_pc):
            loss.backward()          # loss built from several render() calls of the same two Gaussian sets

    A blurry training view back-propagates through 9 render() calls; autograd then adds 9 x 13 per-leaf gradient
    tensors into .grad one launch at a time (124 add_ kernels, 0.7 ms of device time per view at 300 k splats).
    Inside the context the prep backward kernel of every render adds its leaf gradients straight into this sink's
    buffers (the first one writes them), and on exit they become the leaves' .grad (or are added to an existing one)
    -- the same sums in the same order.  Only renders whose inputs are exactly the sets' leaf tensors use the sink.

    When every leaf already HAS a float32 contiguous .grad on entry (the views of a distributed.FlatGradients buffer
    after zero()), those tensors are the sink's buffers: the kernels add into .grad itself and nothing is left to do
    on exit.  The colour decoder's two weight matrices (dyn_pc.rgbdecoder) are handled the same way by the decoder's
    backward (one fixed-order reduction per render that adds into the buffer instead of 2 add_ launches per render),
    and `extra` lists further parameters (the BLCE module's) whose custom backward may add into an existing .grad
    with ONE multi-tensor launch (see `add_into_grads`)."""

    def __init__(self, stat_pc, dyn_pc, extra=()):
        self.leaves = (stat_pc._xyz, stat_pc._scaling, stat_pc._rotation, stat_pc._opacity, stat_pc._features_dc,
                       stat_pc._features_t, dyn_pc.get_control_xyz, dyn_pc._scaling, dyn_pc._rotation, dyn_pc._omega,
                       dyn_pc._opacity, dyn_pc._features_dc, dyn_pc._features_t)
        dec = getattr(dyn_pc, "rgbdecoder", None)
        self.dec_leaves = (dec.mlp1.weight, dec.mlp2.weight) if dec is not None else ()
        self.extra_ids = {id(p) for p in extra}
        self._extra_keep = tuple(extra)  # keeps the ids valid
        self.buffers = None
        self.dec_buffers = None
        self.direct = False
        self.dec_direct = False
        self._prev = None

    def accepts(self, inputs) -> bool:
        return all(a is b and a.is_leaf and a.requires_grad for a, b in zip(inputs, self.leaves))

    @staticmethod
    def _target(p):
        """Where a leaf's gradient lives: its fp32 master when it is a half-stored attribute with one
        (GaussianParams.enable_fp32_masters), else the leaf itself."""
        m = getattr(p, "master", None)
        return p if m is None else m

    @staticmethod
    def _addable(p) -> bool:
        g = LeafGradSink._target(p).grad
        return g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape \
            and not g.requires_grad

    def __enter__(self):
        global _active_sink
        self._prev, _active_sink = _active_sink, self
        if all(self._addable(p) for p in self.leaves):
            self.buffers = {name: self._target(p).grad for p, name in zip(self.leaves, _LEAF_NAMES)}
            self.direct = True
        if self.dec_leaves and all(self._addable(p) for p in self.dec_leaves):
            self.dec_buffers = [p.grad for p in self.dec_leaves]
            self.dec_direct = True
        return self

    def decoder_buffers(self, w1, w2):
        """-> (g_w1, g_w2, accumulate) for a decoder backward whose weights are this sink's, else None."""
        if len(self.dec_leaves) != 2 or w1 is not self.dec_leaves[0] or w2 is not self.dec_leaves[1] or \
                not (w1.requires_grad and w2.requires_grad):
            return None
        if self.dec_buffers is None:
            self.dec_buffers = [torch.empty(p.shape, dtype=torch.float32, device=p.device) for p in self.dec_leaves]
            return self.dec_buffers[0], self.dec_buffers[1], 0
        return self.dec_buffers[0], self.dec_buffers[1], 1

    def add_into_grads(self, params, grads) -> bool:
        """Custom backward of `params` (all listed in `extra`, all with a float32 .grad): .grad += grads in one
        multi-tensor launch; the caller then returns None for them.  False: not applicable, return them normally."""
        if not params or not all(id(p) in self.extra_ids and self._addable(p) for p in params):
            return False
        torch._foreach_add_([p.grad for p in params], list(grads))
        return True

    def __exit__(self, *exc):
        global _active_sink
        _active_sink = self._prev
        if exc[0] is not None:
            return False
        if self.buffers is not None and not self.direct:
            for p, name in zip(self.leaves, _LEAF_NAMES):
                g = self.buffers[name].view_as(p)
                p = self._target(p)     # a half-stored leaf with an fp32 master: the gradient stays fp32, on the master
                if g.dtype != p.dtype:  # fp16 attribute storage without masters: handed over in the leaf's dtype
                    g = g.to(p.dtype)
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
        if self.dec_buffers is not None and not self.dec_direct:
            for p, g in zip(self.dec_leaves, self.dec_buffers):
                g = g.to(p.dtype)
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
        return False


def active_sink():
    return _active_sink


class PrepSplats(torch.autograd.Function):
    """Leaves of the static + dynamic Gaussian sets -> concatenated operator-level inputs at one time instant
    (times [2]) or at K instants in one launch (times [K,2]: means [K,N,3], quats [K,N,4], colors [K,N,9]; scales and
    opacities do not depend on time and come once).

    Replaces /root/reference/gaussian_renderer/__init__.py:69-125,181-185 (see csrc/prep.hip)."""

    @staticmethod
    def forward(ctx, times, s_xyz, s_scaling, s_rotation, s_opacity, s_fdc, s_ft, d_control, d_ncp, d_scaling,
                d_rotation, d_omega, d_opacity, d_fdc, d_ft, d_trbf):
        lib = _lib.load()
        # the caller's tensor objects (a LeafGradSink recognises its leaves by identity)
        ctx.leaf_inputs = (s_xyz, s_scaling, s_rotation, s_opacity, s_fdc, s_ft, d_control, d_scaling, d_rotation,
                           d_omega, d_opacity, d_fdc, d_ft)
        # fp16 attribute storage (BASELINE config #5): when every attribute leaf is float16 the kernel reads the halves
        # directly; positions, control points and time centres are always fp32
        attrs = (s_scaling, s_rotation, s_opacity, s_fdc, s_ft, d_scaling, d_rotation, d_omega, d_opacity, d_fdc, d_ft)
        half = all(a.dtype == torch.float16 for a in attrs)
        ctx.attr_dtypes = tuple(a.dtype for a in attrs)
        ctx.half = half
        ctx.sizes = (s_xyz.shape[0], d_control.shape[0])
        F = _fast.get()
        if F is not None:  # the same body in C++ (csrc/fastpath.cpp)
            (means, quats, scales, opac, colors, times, d_ncp, d_trbf, n_conv) = F.prep_fwd(
                times, s_xyz, d_control, d_ncp, d_trbf, list(attrs), half, stream_int())
            _lib.attr_conversions += n_conv
            ctx.save_for_backward(times, d_ncp, d_trbf, scales, opac)
            return means, quats, scales, opac, colors
        times, s_xyz, d_control, d_trbf = map(f32c, (times, s_xyz, d_control, d_trbf))
        (s_scaling, s_rotation, s_opacity, s_fdc, s_ft, d_scaling, d_rotation, d_omega, d_opacity, d_fdc,
         d_ft) = (attr_c(a, half) for a in attrs)
        d_ncp = d_ncp.to(torch.int64).contiguous()
        Ns, Nd = s_xyz.shape[0], d_control.shape[0]
        N = Ns + Nd
        dev = times.device
        # times [2]: one instant; times [K,2]: K instants in one launch, means / quats / colors get a leading K
        lead = (times.shape[0],) if times.dim() == 2 else ()
        K = lead[0] if lead else 1
        means = torch.empty(*lead, N, 3, dtype=torch.float32, device=dev)
        quats = torch.empty(*lead, N, 4, dtype=torch.float32, device=dev)
        scales = torch.empty(N, 3, dtype=torch.float32, device=dev)
        opac = torch.empty(N, dtype=torch.float32, device=dev)
        colors = torch.empty(*lead, N, 9, dtype=torch.float32, device=dev)
        fwd = lib.mobgs_prep_fwd_many_f16 if half else lib.mobgs_prep_fwd_many
        check(fwd(K, Ns, Nd, ptr(times), ptr(s_xyz), ptr(s_scaling), ptr(s_rotation), ptr(s_opacity), ptr(s_fdc),
                  ptr(s_ft), ptr(d_control), ptr(d_ncp), ptr(d_scaling), ptr(d_rotation), ptr(d_omega),
                  ptr(d_opacity), ptr(d_fdc), ptr(d_ft), ptr(d_trbf), ptr(means), ptr(quats), ptr(scales), ptr(opac),
                  ptr(colors), stream()), "mobgs_prep_fwd")
        ctx.save_for_backward(times, d_ncp, d_trbf, scales, opac)
        return means, quats, scales, opac, colors

    @staticmethod
    def backward(ctx, v_means, v_quats, v_scales, v_opac, v_colors):
        lib = _lib.load()
        times, d_ncp, d_trbf, scales, opac = ctx.saved_tensors
        Ns, Nd = ctx.sizes
        dev = times.device

        sink = _active_sink
        use_sink = sink is not None and getattr(ctx, "leaf_inputs", None) is not None and sink.accepts(ctx.leaf_inputs)
        # attribute gradients: fp32 in the sink's accumulation buffers (cast once when the context exits), otherwise
        # the dtype of the leaves (half leaves need half .grad tensors: written as such by the kernel)
        g_half = ctx.half and not use_sink

        def E(*shape, attr=False):
            return torch.empty(*shape, dtype=torch.float16 if (attr and g_half) else torch.float32, device=dev)

        accumulate = 0
        F = _fast.get()
        if F is not None:
            have = use_sink and sink.buffers is not None
            out = F.prep_bwd(Ns, Nd, times, d_ncp, d_trbf, scales, opac, v_means, v_quats, v_scales, v_opac, v_colors,
                             [sink.buffers[n_] for n_ in _LEAF_NAMES] if have else [], g_half, 1 if have else 0,
                             stream_int())
            g = sink.buffers if have else dict(zip(_LEAF_NAMES, out))
            if use_sink:
                sink.buffers = g
        elif use_sink and sink.buffers is not None:
            g, accumulate = sink.buffers, 1
        else:
            g = {"s_xyz": E(Ns, 3), "s_scaling": E(Ns, 3, attr=True), "s_rotation": E(Ns, 4, attr=True),
                 "s_opacity": E(Ns, 1, attr=True), "s_fdc": E(Ns, 6, attr=True), "s_ft": E(Ns, 3, attr=True),
                 "d_control": E(Nd, 12, 3), "d_scaling": E(Nd, 3, attr=True), "d_rotation": E(Nd, 4, attr=True),
                 "d_omega": E(Nd, 4, attr=True), "d_opacity": E(Nd, 1, attr=True), "d_fdc": E(Nd, 6, attr=True),
                 "d_ft": E(Nd, 3, attr=True)}
            if use_sink:
                sink.buffers = g
        if F is None:
            c = [f32c(v) if v is not None else None for v in (v_means, v_quats, v_scales, v_opac, v_colors)]
            bwd = lib.mobgs_prep_bwd_many_f16 if g_half else lib.mobgs_prep_bwd_many
            check(bwd(times.shape[0] if times.dim() == 2 else 1, Ns, Nd, ptr(times), ptr(d_ncp), ptr(d_trbf), ptr(scales), ptr(opac), ptr(c[0]), ptr(c[1]),
                      ptr(c[2]), ptr(c[3]), ptr(c[4]), ptr(g["s_xyz"]), ptr(g["s_scaling"]), ptr(g["s_rotation"]),
                      ptr(g["s_opacity"]), ptr(g["s_fdc"]), ptr(g["s_ft"]), ptr(g["d_control"]), ptr(g["d_scaling"]),
                      ptr(g["d_rotation"]), ptr(g["d_omega"]), ptr(g["d_opacity"]), ptr(g["d_fdc"]), ptr(g["d_ft"]),
                      accumulate, stream()), "mobgs_prep_bwd")
        if not use_sink and not g_half:  # mixed / other dtypes: autograd wants the leaf's dtype back
            names = ("s_scaling", "s_rotation", "s_opacity", "s_fdc", "s_ft", "d_scaling", "d_rotation", "d_omega",
                     "d_opacity", "d_fdc", "d_ft")
            for n_, dt in zip(names, ctx.attr_dtypes):
                if dt != torch.float32:
                    g[n_] = g[n_].to(dt)
        if use_sink:
            return (None,) * 16
        return (None, g["s_xyz"], g["s_scaling"], g["s_rotation"], g["s_opacity"], g["s_fdc"], g["s_ft"],
                g["d_control"], None, g["d_scaling"], g["d_rotation"], g["d_omega"], g["d_opacity"], g["d_fdc"],
                g["d_ft"], None)


def _decoder_strides(C, P, rays, intr, c2w):
    """Floats between consecutive images' ray maps / intrinsics / poses of a batch of C images (0 = shared)."""
    if C <= 1:
        return 0, 0, 0
    rs = 6 * P if (rays is not None and rays.numel() == C * 6 * P) else 0
    is_ = 4 if (intr is not None and intr.numel() == 4 * C) else 0
    cs = c2w.numel() // C if (c2w is not None and c2w.dim() == 3 and c2w.shape[0] == C) else 0
    return rs, is_, cs


class Decode(torch.autograd.Function):
    """Channels-last compositor image (+alpha) -> planar rgb [3,H,W] (+ expected depth [H,W]).
    Rays: either the reference's map `rays` [6,H,W], or (rays=None) the pinhole parameters `intr` = [fx,fy,cx,cy]
    and `c2w` [3,4] from which the kernel generates them (gradient flows into c2w)."""

    @staticmethod
    def forward(ctx, feat_hw, alphas, rays, intr, c2w, w1, w2, has_depth: bool, pre_rgb=None, pre_depth=None, _chan=None):
        """pre_rgb / pre_depth: this node's outputs, already computed by the forward compositor's decoder epilogue
        (rendering.SharedProjection.composite_decode) -- nothing is launched here, the backward pass is unchanged.
        _chan = (c0, n) (DecodeWithChannels, host fast path): the kernel also hands out channels [c0, c0 + n) of the image
        as a contiguous tensor, left in ctx.chan_out."""
        lib = _lib.load()
        ctx.set_materialize_grads(False)  # an unused depth output costs no zero image in backward
        ctx.w_inputs = (w1, w2)  # the caller's tensor objects (a LeafGradSink recognises its weights by identity)
        feat_hw, w1, w2 = map(f32c, (feat_hw, w1, w2))
        H, W, CF = feat_hw.shape[-3:]
        P = H * W
        dev = feat_hw.device
        alphas_c = f32c(alphas) if alphas is not None else None
        rays_c = f32c(rays) if rays is not None else None
        intr_c = c2w_c = None
        C = feat_hw.numel() // (P * CF)   # > 1: a batch [C,H,W,CF] decoded in one launch -> rgb [C,3,H,W]
        if rays_c is None:  # pinhole parameters: the kernel reads the 4 intrinsics and the first 12 pose entries
            intr_c, c2w_c = f32c(intr.detach()), f32c(c2w.detach())
            shared = intr_c.numel() == 4 and c2w_c.numel() in (12, 16) and c2w_c.dim() == 2
            per_img = C > 1 and intr_c.numel() in (4, 4 * C) and c2w_c.dim() == 3 and c2w_c.shape[0] == C and \
                c2w_c.numel() in (12 * C, 16 * C)
            if not (shared or per_img):
                raise ValueError("decode: intr must be [fx, fy, cx, cy] and c2w a [3,4] or [4,4] camera-to-world matrix "
                                 "(a batch of C images: [C,4] / [C,3|4,4], or shared ones)")
        F = _fast.get()
        if pre_rgb is not None:
            rgb, depth = pre_rgb.view_as(pre_rgb), (pre_depth.view_as(pre_depth) if has_depth else None)
        elif F is not None:
            rgb, depth, ctx.chan_out = F.decoder_fwd(H, W, CF, bool(has_depth), feat_hw, alphas_c, rays_c, intr_c, c2w_c,
                                                     w1, w2, stream_int(), _chan[0] if _chan else 0,
                                                     _chan[1] if _chan else 0)
        else:
            lead = (C,) if feat_hw.dim() == 4 else ()
            rgb = torch.empty(*lead, 3, H, W, dtype=torch.float32, device=dev)
            depth = torch.empty(*lead, H, W, dtype=torch.float32, device=dev) if has_depth else None
            rs, is_, cs = _decoder_strides(C, P, rays_c, intr_c, c2w_c)
            check(lib.mobgs_decoder_fwd_many(C, P, CF, int(has_depth), W, ptr(feat_hw), ptr(alphas_c), ptr(rays_c), rs,
                                             ptr(intr_c), is_, ptr(c2w_c), cs, ptr(w1), ptr(w2), ptr(rgb), ptr(depth),
                                             stream()), "mobgs_decoder_fwd")
        ctx.save_for_backward(feat_hw, alphas_c, rays_c, intr_c, c2w_c, w1, w2)
        ctx.has_depth = has_depth
        ctx.rays_need_grad = rays is not None and ctx.needs_input_grad[2]
        ctx.c2w_needs_grad = rays is None and ctx.needs_input_grad[4]
        ctx.feat_shape = feat_hw.shape
        if has_depth:
            return rgb, depth
        return rgb, rgb.new_empty(0)

    @staticmethod
    def backward(ctx, v_rgb, v_depth, _v_chan=None, _chan_c0=0):
        """_v_chan / _chan_c0 (DecodeWithChannels, host fast path): the cotangent of the channels handed out by forward;
        the kernel writes it into those channels of the image's gradient."""
        lib = _lib.load()
        feat_hw, alphas, rays, intr, c2w, w1, w2 = ctx.saved_tensors
        if v_rgb is None and v_depth is None:
            return (None,) * 10
        H, W, CF = feat_hw.shape[-3:]
        P = H * W
        dev = feat_hw.device
        has_depth = ctx.has_depth
        sunk = _active_sink.decoder_buffers(*ctx.w_inputs) if _active_sink is not None else None
        F = _fast.get()
        if F is not None:
            v_feat, v_alphas, v_rays, g_c2w, g_w1, g_w2 = F.decoder_bwd(
                H, W, CF, bool(has_depth), feat_hw, alphas, rays, intr, c2w, w1, w2, v_rgb, v_depth,
                list(ctx.feat_shape), bool(ctx.rays_need_grad), bool(ctx.c2w_needs_grad),
                sunk[0] if sunk is not None else None, sunk[1] if sunk is not None else None,
                sunk[2] if sunk is not None else 0, stream_int(), _v_chan, int(_chan_c0))
            if sunk is not None:
                return v_feat, v_alphas, v_rays, None, g_c2w, None, None, None, None, None
            return v_feat, v_alphas, v_rays, None, g_c2w, g_w1, g_w2, None, None, None
        C = feat_hw.numel() // (P * CF)
        v_rgb = f32c(v_rgb) if v_rgb is not None else torch.zeros(C, 3, H, W, dtype=torch.float32, device=dev)
        v_depth = f32c(v_depth) if (has_depth and v_depth is not None) else None
        v_feat = torch.empty(ctx.feat_shape, dtype=torch.float32, device=dev)
        v_alphas = torch.empty(alphas.shape, dtype=torch.float32, device=dev) if has_depth else None
        v_rays = torch.empty_like(rays) if ctx.rays_need_grad else None
        g_c2w = torch.empty_like(c2w) if ctx.c2w_needs_grad else None  # [3,4] or [4,4] (per image) like the input
        nb = lib.mobgs_decoder_bwd_blocks(P)
        partial = torch.empty(C * nb, 102, dtype=torch.float32, device=dev)
        if sunk is not None:
            g_w1, g_w2, accumulate = sunk
        else:
            g_w1, g_w2, accumulate = torch.empty_like(w1), torch.empty_like(w2), 0
        rs, is_, cs = _decoder_strides(C, P, rays, intr, c2w)
        check(lib.mobgs_decoder_bwd_many(C, P, CF, int(has_depth), W, ptr(feat_hw), ptr(alphas), ptr(rays), rs,
                                         ptr(intr), is_, ptr(c2w), cs, ptr(w1), ptr(w2), ptr(v_rgb), ptr(v_depth),
                                         ptr(v_feat), ptr(v_alphas), ptr(v_rays), ptr(partial), ptr(g_w1), ptr(g_w2),
                                         ptr(g_c2w), (g_c2w.numel() // (C if cs else 1)) if g_c2w is not None else 0,
                                         accumulate, stream()),
              "mobgs_decoder_bwd")
        if sunk is not None:
            return v_feat, v_alphas, v_rays, None, g_c2w, None, None, None, None, None
        return v_feat, v_alphas, v_rays, None, g_c2w, g_w1, g_w2, None, None, None


class DecodeWithChannels(torch.autograd.Function):
    """Decode (no depth) of an image that carries `n` further channels from `c0` on, which are handed out as a
    tensor of their own [..., n].  Slicing them off the image in PyTorch costs every backward pass a zero image of the
    image's size, a strided copy and an add with the decoder's gradient; here the decoder's gradient buffer (zero in
    the channels it does not read) simply receives their cotangent.  get_flow(): 9 features + 2 flow channels."""

    @staticmethod
    def forward(ctx, feat_hw, alphas, rays, intr, c2w, w1, w2, c0: int, n: int):
        ctx.chan = (int(c0), int(n))
        ctx.chan_out = None
        rgb, _ = Decode.forward(ctx, feat_hw, alphas, rays, intr, c2w, w1, w2, False, None, None,
                                ctx.chan if _fast.get() is not None else None)
        chan, ctx.chan_out = ctx.chan_out, None
        if chan is None:   # (Python host path: the slice as a strided copy)
            chan = feat_hw[..., c0:c0 + n].contiguous()
        return rgb, chan

    @staticmethod
    def backward(ctx, v_rgb, v_chan):
        c0, n = ctx.chan
        if v_rgb is None and v_chan is None:
            return (None,) * 9
        if v_rgb is None:  # only the extra channels were used
            v_feat = torch.zeros(ctx.feat_shape, dtype=torch.float32, device=v_chan.device)
            v_feat[..., c0:c0 + n].copy_(v_chan)
            return (v_feat,) + (None,) * 8
        if _fast.get() is not None and v_chan is not None:
            g = Decode.backward(ctx, v_rgb, None, v_chan, c0)   # (the kernel drops the cotangent into its channels)
            return tuple(g[:7]) + (None, None)
        g = Decode.backward(ctx, v_rgb, None)
        if v_chan is not None:
            g[0][..., c0:c0 + n].copy_(v_chan)
        return tuple(g[:7]) + (None, None)


def _decode_args(feat_hw: Tensor, alphas: Optional[Tensor], rays):
    """Shapes for Decode: one image [H,W,CF] (any number of leading 1-dimensions, as the reference's [1,H,W,CF]), or a
    batch [C,H,W,CF] with C > 1 decoded in ONE launch.  -> (feat, alphas, ray map | None, intr | None, c2w | None)."""
    H, W, CF = feat_hw.shape[-3:]
    C = feat_hw.numel() // (H * W * CF)
    lead = (C,) if C > 1 else ()
    feat = feat_hw.reshape(*lead, H, W, CF)
    if alphas is not None:
        alphas = alphas.reshape(*lead, H, W)
    if isinstance(rays, (tuple, list)):
        intr, c2w = rays
        if C == 1:   # a stacked "batch" of one camera is that camera
            intr = intr.reshape(4)
            c2w = c2w.reshape(c2w.shape[-2:]) if c2w.dim() == 3 else c2w
        if C > 1 and c2w.dim() == 2 and c2w.requires_grad:
            c2w = c2w.expand(C, *c2w.shape).contiguous()   # one gradient per image, summed by autograd
        return feat, alphas, None, intr, c2w
    if C > 1 and rays.numel() == 6 * H * W and rays.requires_grad:
        raise NotImplementedError("decode: a ray map shared by a batch of images cannot receive a gradient")
    return feat, alphas, rays.reshape(*((C,) if rays.numel() == C * 6 * H * W and C > 1 else ()), 6, H, W), None, None


def decode_with_channels(feat_hw: Tensor, alphas: Optional[Tensor], rays, w1: Tensor, w2: Tensor, c0: int, n: int):
    """decode(..., has_depth=False) + feat_hw[..., c0:c0+n] as a tensor of its own (see DecodeWithChannels):
    -> (rgb [3,H,W], channels [H,W,n]); a batch [C,H,W,CF]: ([C,3,H,W], [C,H,W,n])."""
    feat, alphas, ray_map, intr, c2w = _decode_args(feat_hw, alphas, rays)
    return DecodeWithChannels.apply(feat, alphas, ray_map, intr, c2w, w1, w2, c0, n)


def decode(feat_hw: Tensor, alphas: Optional[Tensor], rays, w1: Tensor, w2: Tensor, has_depth: bool):
    """feat_hw [..,H,W,CF>=9(+1)], alphas [..,H,W] or [..,H,W,1] -> rgb [3,H,W], depth [H,W]|None; a batch of C > 1 images
    [C,H,W,CF] is decoded in one launch -> rgb [C,3,H,W], depth [C,H,W]|None.
    `rays`: the reference's cam_ray map [1,6,H,W] (batch: shared, or [C,6,H,W]), or a pair (intr [4] = fx,fy,cx,cy,
    c2w [3,4] or [4,4]) to have the kernel generate the pinhole rays itself (batch: shared, or [C,4] / [C,3|4,4])."""
    feat, alphas, ray_map, intr, c2w = _decode_args(feat_hw, alphas, rays)
    rgb, depth = Decode.apply(feat, alphas, ray_map, intr, c2w, w1, w2, bool(has_depth))
    return rgb, (depth if has_depth else None)


def decode_nchw(feat: Tensor, rays: Tensor, w1: Tensor, w2: Tensor) -> Tensor:
    """Sandwich.forward signature: feat [1,9,H,W], rays [1,6,H,W] -> [1,3,H,W]."""
    if feat.shape[0] != 1:
        raise NotImplementedError("decoder batch size must be 1 (as in every reference call)")
    rgb, _ = decode(feat[0].permute(1, 2, 0), None, rays, w1, w2, False)
    return rgb.unsqueeze(0)
