"""Render API (boundary B1): drop-in for the reference's `gaussian_renderer` package.

    from mobgs_amd.gaussian_renderer import render, get_flow, get_flow_static

mirror /root/reference/gaussian_renderer/__init__.py:59-316 (render), :318-492 (get_flow), :494-552
(get_flow_static): same positional/keyword arguments, same 22-key result dict (including the `s_depth` quirk
of :250), same autograd contract (`viewspace_points` is a non-leaf with retain_grad(), gradients reach every
Gaussian leaf, the decoder weights, `w2c` / the camera's world_view_transform and cam_ray).

Everything per-Gaussian and per-pixel runs in libmobgs_hip.so:
    prep (Hermite spline, rotation, activations, colour features, static|dynamic concat)  -> csrc/prep.hip
    projection / tile lists / compositing                                                 -> mobgs_amd.rendering
    expected-depth normalisation + Sandwich decoder                                       -> csrc/decoder.hip
Unused reference kwargs (pipe, scaling_modifier, override_color, stage, cam_type, ...) are accepted and ignored,
exactly as the reference ignores them.
"""
from __future__ import annotations

import weakref

import torch

from .. import rendering as _R
from .._lib import DerivedCache
from ..ops import PrepSplats, decode, decode_with_channels
from . import network_gui  # noqa: F401  (train.py imports it from here)

__all__ = ["render", "render_many", "viewspace_grad", "get_flow", "get_flow_many", "get_flow_static",
           "interpolate_cubic_hermite", "network_gui"]

# True: the static-only / dynamic-only images of a train-mode render() come from one layered compositing pass over
# the lists of the combined render (csrc/raster_layers.hip); False: one rasterization per set, call for call like
# the reference (kept for A/B tests)
FUSE_LAYERS = True
# True: those auxiliary images (s_render, s_alpha, d_render, d_depth, d_alpha) are computed on first access of
# their dict entry.  train.py asks every one of the 8 latent sub-frame renders of a blurry view for them
# (:512-516) and then reads only "render"/"depth" -- with lazy entries those calls cost a lean render.
LAZY_AUX = True

_PENDING = object()


class RenderResult(dict):
    """The reference's result dict; entries registered with defer() are materialised on first access (each defer()
    call registers one group of keys computed together by `thunk() -> {key: value}`)."""

    def defer(self, keys, thunk):
        if not hasattr(self, "_thunks"):
            self._thunks = {}
        for k in keys:
            dict.__setitem__(self, k, _PENDING)
            self._thunks[k] = thunk

    def _materialise(self, key=None):
        thunks = getattr(self, "_thunks", None)
        if not thunks:
            return
        todo = [thunks[key]] if key is not None else list({id(t): t for t in thunks.values()}.values())
        for thunk in todo:
            for k in [k for k, t in thunks.items() if t is thunk]:
                del thunks[k]
            dict.update(self, thunk())

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if v is _PENDING:
            self._materialise(key)
            v = dict.__getitem__(self, key)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        self._materialise()
        return dict.items(self)

    def values(self):
        self._materialise()
        return dict.values(self)

    def copy(self):
        self._materialise()
        return dict(self)

    # dict(out), {**out}, out | other go through PyDict_Merge, which reads the storage of a dict SUBCLASS directly
    # unless the subclass overrides __iter__ (then it uses keys() + __getitem__): overriding it is what keeps the
    # pending marker from leaking through those paths.
    def __iter__(self):
        return dict.__iter__(self)

    def keys(self):
        return dict.keys(self)

    def pop(self, key, *default):
        if key in self:
            self[key]  # materialise its group first
        return dict.pop(self, key, *default)

    def setdefault(self, key, default=None):
        return self[key] if key in self else dict.setdefault(self, key, default)

    def __reduce__(self):  # pickling / copy.copy / copy.deepcopy: a plain dict of materialised values
        self._materialise()
        return (dict, (dict(dict.items(self)),))


def _device_of(pc):
    return pc.get_xyz.device


_time_cache = {}


_subframe_times = {}


def _times(cam, delta_exposure, dev):
    """[t_feat, t_curve] on the device.  No host->device copy on the hot path (a pageable H2D copy blocks the
    host until the stream drains, which serialises consecutive render() calls): constants are cached per value,
    a device-resident delta_exposure (BLCE exposure offset) is combined with device ops only."""
    def const(t):
        key = (str(dev), float(t))
        v = _time_cache.get(key)
        if v is None:
            if len(_time_cache) > 4096:
                _time_cache.clear()
            v = torch.tensor([float(t), min(max(float(t), 0.0), 1.0)], dtype=torch.float32, device=dev)
            _time_cache[key] = v
        return v

    st = getattr(cam, "static_times", None)
    if st is not None and delta_exposure is None:
        # a camera whose time lives in a device buffer [t, clamp(t, 0, 1)] that its owner rewrites in place
        # (mobgs_amd.graphed: a captured HIP graph must read the time from memory, not from a constant baked at capture)
        return st
    if delta_exposure is None:
        return const(cam.time)
    if torch.is_tensor(delta_exposure) and delta_exposure.is_cuda:
        # train.py hands over exposure_time[k], an element of the view's vector of K offsets: the K time pairs are
        # computed with ONE set of (4) launches on the whole vector and the element's row is picked from the table
        base = delta_exposure._base
        if base is not None and base.dim() == 1 and delta_exposure.dim() == 0 and base.is_contiguous():
            key = (id(base), float(cam.time), float(cam.max_time))
            hit = _subframe_times.get(key)
            if hit is None or hit[0]() is not base or hit[1] != base._version:
                if len(_subframe_times) > 64:
                    _subframe_times.clear()
                t = const(cam.time)[0] + base.detach().to(torch.float32) / cam.max_time
                hit = (weakref.ref(base), base._version, torch.stack([t, torch.clamp(t, 0.0, 1.0)], dim=1))
                _subframe_times[key] = hit
            return hit[2][delta_exposure.storage_offset() - base.storage_offset()]
        t = const(cam.time)[0] + delta_exposure.detach().to(torch.float32).reshape(()) / cam.max_time
        return torch.stack([t, torch.clamp(t, 0.0, 1.0)])
    d = float(delta_exposure)
    # same float32 arithmetic as the reference's tensor expression time + delta / max_time
    t32 = torch.tensor(float(cam.time), dtype=torch.float32) + torch.tensor(d, dtype=torch.float32) / cam.max_time
    return const(float(t32))


# Enumeration order of the binning (rendering.SharedProjection(order=), include/mobgs_hip.h enum_order): a Morton order of
# the splats' 3-D positions, cached per (static set, dynamic set) and recomputed every ENUM_ORDER_REFRESH calls (~0.5 ms of small torch launches and a sort: amortised to a fraction of a microsecond per call; the
# Gaussians move slowly; densification changes N and forces a new one).  A performance structure only: lists, images and
# gradients are bit-identical with any order or none (tests/test_gpu_fused_lists.py).
ENUM_ORDER = __import__("os").environ.get("MOBGS_ENUM_ORDER", "1") != "0"
ENUM_ORDER_REFRESH = 2048
_enum_cache = {}


def _rows_coherent(pc):
    return getattr(pc, "rows_coherent", -1) == pc.get_xyz.shape[0]


def _enum_order(stat_pc, dyn_pc, means, cameras=1):
    if not (ENUM_ORDER and _R.FUSED_LISTS):
        return None
    if _rows_coherent(stat_pc) and _rows_coherent(dyn_pc):
        return _R.COHERENT   # both sets store their rows along a Morton curve (GaussianParams.spatial_sort_): no order needed
    key = (id(stat_pc), id(dyn_pc))
    if means is None:   # (the fused-prep path has no activated positions yet: the raw ones order just as well)
        class _Raw:
            shape = (stat_pc._xyz.shape[0] + dyn_pc.get_control_xyz.shape[0], 3)
            device = stat_pc._xyz.device

            @staticmethod
            def dim():
                return 2
        means = _Raw
    n = int(means.shape[-2])
    e = _enum_cache.get(key)
    fresh = e is None or e["stat"]() is not stat_pc or e["dyn"]() is not dyn_pc or e["n"] != n or \
        e["dev"] != means.device
    if not fresh:
        e["calls"] += 1
        fresh = e["calls"] >= ENUM_ORDER_REFRESH
    if fresh:
        if torch.cuda.is_current_stream_capturing():   # (a sort allocates: not inside a HIP-graph capture)
            return None if e is None or e["n"] != n or e["dev"] != means.device else e["by_c"].get(cameras)
        if len(_enum_cache) > 16:
            _enum_cache.clear()
        if not torch.is_tensor(means):
            means = torch.cat((stat_pc._xyz.detach().float(),
                               dyn_pc.get_control_xyz.detach().float().mean(1) * 1e-2), 0)
        base = _R.spatial_order(means if means.dim() == 2 else means[0])
        e = _enum_cache[key] = {"stat": weakref.ref(stat_pc), "dyn": weakref.ref(dyn_pc), "n": n, "dev": means.device,
                                "calls": 0, "by_c": {1: base}}
    order = e["by_c"].get(cameras)
    if order is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        base = e["by_c"][1]
        order = e["by_c"][cameras] = (base[None, :] + n * torch.arange(cameras, device=base.device,
                                                                        dtype=torch.int32)[:, None]).reshape(-1).contiguous()
    return order


def _raw_inputs(stat_pc, dyn_pc):
    """The 15 tensors ops.PrepSplats takes behind `times`."""
    return (stat_pc._xyz, stat_pc._scaling, stat_pc._rotation, stat_pc._opacity, stat_pc._features_dc,
            stat_pc._features_t, dyn_pc.get_control_xyz, dyn_pc.current_control_num, dyn_pc._scaling, dyn_pc._rotation,
            dyn_pc._omega, dyn_pc._opacity, dyn_pc._features_dc, dyn_pc._features_t, dyn_pc.get_trbfcenter)


# Lean render(): the per-splat state (spline position, activations, colour features) is built INSIDE the projection kernel
# instead of by a launch of its own (rendering._PrepProjectAndBin).  Same arithmetic, same outputs,
# same gradients; MOBGS_FUSE_PREP=0 keeps the two launches (A/B).  Train-mode renders take the same path (their static /
# dynamic layers are class passes over the same packed records); the `coherent` offset, half-stored attributes and the
# python host path keep the two launches.
FUSE_PREP = __import__("os").environ.get("MOBGS_FUSE_PREP", "1") != "0"


def _can_fuse_prep(raw, times):
    from .. import _fast
    if not (FUSE_PREP and _R.SPECULATIVE_BINNING and _fast.get() is not None and times.dim() == 1):
        return False
    return all(t.dtype == torch.float32 for i, t in enumerate(raw) if i != 7) and raw[0].shape[0] + raw[6].shape[0] > 0


def _prep(stat_pc, dyn_pc, times):
    return PrepSplats.apply(times, stat_pc._xyz, stat_pc._scaling, stat_pc._rotation, stat_pc._opacity,
                            stat_pc._features_dc, stat_pc._features_t, dyn_pc.get_control_xyz,
                            dyn_pc.current_control_num, dyn_pc._scaling, dyn_pc._rotation, dyn_pc._omega,
                            dyn_pc._opacity, dyn_pc._features_dc, dyn_pc._features_t, dyn_pc.get_trbfcenter)


# True: cameras that expose their pinhole parameters (`ray_intrinsics`, `ray_c2w` -- mobgs_amd.camera.PinholeCamera,
# mobgs_amd.blce.WarpedCamera) get their rays generated inside the decoder kernel; other cameras (e.g. the
# reference's scene.cameras.Camera) are read through their `cam_ray` map exactly as the reference does
INKERNEL_RAYS = True


def _rays_of(cam):
    if INKERNEL_RAYS and hasattr(cam, "ray_c2w") and hasattr(cam, "ray_intrinsics"):
        return (cam.ray_intrinsics, cam.ray_c2w)
    return cam.cam_ray


def _rays_of_many(cams):
    """(intr [K,4], c2w [K,3|4,4]) for a batched decode of K cameras that all expose their pinhole parameters with
    poses of one shape; else None (the caller decodes image by image)."""
    if not (INKERNEL_RAYS and all(hasattr(c, "ray_c2w") and hasattr(c, "ray_intrinsics") for c in cams)):
        return None
    poses = [c.ray_c2w for c in cams]
    if len({tuple(p.shape) for p in poses}) != 1:
        return None
    return torch.stack([c.ray_intrinsics for c in cams]), torch.stack(poses)


_bg9_cache = DerivedCache()


def _bg9(bg_color):
    """The 9-channel background row the reference builds per call (:77), bg_color[:3] three times, as [1,9]."""
    return _bg9_cache.get((bg_color,), lambda: torch.cat([bg_color[:3]] * 3, dim=-1)[None])


_bg11_cache = DerivedCache()


def _bg11(bg1):
    """[1,9] feature background + zeros for the two flow channels splatted in the same pass."""
    return _bg11_cache.get((bg1,), lambda: torch.cat([bg1, bg1.new_zeros(1, 2)], dim=-1))


def _decoder_weights(dyn_pc):
    dec = dyn_pc.rgbdecoder
    return dec.mlp1.weight, dec.mlp2.weight


def interpolate_cubic_hermite(signal, times, N):
    """API of /root/reference/gaussian_renderer/__init__.py:23-56: signal [Nd,3,K], times [Nd,3,1], N [Nd,1].
    (Convenience wrapper; render() evaluates the spline inside the fused prep kernel.)"""
    ctrl = signal.permute(0, 2, 1).contiguous()
    nd = ctrl.shape[0]
    if ctrl.shape[1] != 12:
        raise NotImplementedError("the fused spline kernel is built for 12 control points (control_num = 12)")
    dev = ctrl.device
    t = times.reshape(nd, -1)[0, 0].to(torch.float32)
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)  # noqa: E731
    means, _, _, _, _ = PrepSplats.apply(torch.stack([t, t]), z(0, 3), z(0, 3), z(0, 4), z(0, 1), z(0, 6), z(0, 3),
                                         ctrl, N, z(nd, 3), z(nd, 4), z(nd, 4), z(nd, 1), z(nd, 6), z(nd, 3),
                                         z(nd, 1))
    return means * 1e2


def _keep_grad(t: torch.Tensor) -> None:
    """`viewspace_points.retain_grad()` of the reference (gaussian_renderer/__init__.py:121-124) without the copy:
    retain_grad() clones the incoming gradient (2.4 MB device copy per render); the hook keeps a reference to the
    gradient tensor itself, which nothing downstream writes to."""
    if not t.requires_grad:
        return
    ref = weakref.ref(t)
    seen = []  # reading .grad of a non-leaf that has none yet warns: remember whether a pass has stored one

    def hook(g):
        target = ref()
        if target is not None:
            # accumulate across backward passes like retain_grad() does (train.py calls photo_loss.backward(
            # retain_graph=True) and then loss.backward(), :629,:678); the first pass keeps the tensor itself
            target.grad = target.grad + g if seen else g
            seen.append(True)

    t.register_hook(hook)


def _scoped(fn, stat_at=1):
    """The entry point runs inside rendering.hint_scope(token of its (stat_pc, dyn_pc) pair): arena capacities, list-length
    hints and key-segment strides are kept per scene (and per thread), not per (device, N, W, H) alone."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        stat_pc = a[stat_at] if len(a) > stat_at else k.get("stat_pc")
        dyn_pc = a[stat_at + 1] if len(a) > stat_at + 1 else k.get("dyn_pc")
        with _R.hint_scope(_R.scene_token(stat_pc, dyn_pc)):
            return fn(*a, **k)
    return wrapped


@_scoped
def render(viewpoint_camera, stat_pc, dyn_pc, pipe, bg_color, scaling_modifier=1.0, override_color=None,
           stage="fine", cam_type=None, is_static=False, over_t=None, over_vde=None, get_static=False,
           get_dynamic=False, stat_stat=True, ref_wc=None, iter_fact=1, flow=None, coherent=None, target_ts=None,
           target_w2cs=None, get_heatmap=False, w2c=None, delta_exposure=None, get_flow=False, cluster=None):
    """Render the scene (see module docstring).  Returns the reference's result dict."""
    if cluster is not None:
        raise NameError("name 'labels' is not defined")  # the reference raises exactly this (:314-315)
    cam = viewpoint_camera
    dev = _device_of(dyn_pc)
    W, H = int(cam.image_width), int(cam.image_height)
    viewmat = cam.world_view_transform.transpose(0, 1) if w2c is None else w2c
    K = cam.K
    bg1 = _bg9(bg_color)
    bg = bg1[0]
    w1, w2 = _decoder_weights(dyn_pc)
    Ns = stat_pc.get_xyz.shape[0]

    dyn_sl, stat_sl, all_sl = slice(Ns, None), slice(0, Ns), slice(None)
    times = _times(cam, delta_exposure, dev)
    raw = _raw_inputs(stat_pc, dyn_pc)
    # (train-mode renders too: their static / dynamic layers are class passes over the SAME packed records; the layered
    # walk that reads the colour features as an array gets them from a prep launch of its own, see aux_images)
    fuse_prep = (coherent is None and not (delta_exposure is not None and get_flow) and _can_fuse_prep(raw, times)
                 and ((not get_static and not get_dynamic) or (FUSE_LAYERS and _R.CLASS_PASSES)))
    sp = None
    if _R.path_log is not None:   # (tests: which prep path this call takes)
        _R.path_log.append({"dir": "prep", "D": 0, "fused": bool(fuse_prep)})
    if fuse_prep:
        # (cols: the token the compositing node returns its colour gradient through -- never read as data)
        sp, means, quats, scales, opac = _R.SharedProjection.from_raw(times, raw, viewmat[None], K[None], W, H,
                                                                      order=_enum_order(stat_pc, dyn_pc, None))
        cols = sp.state_colors
    else:
        means, quats, scales, opac, cols = _prep(stat_pc, dyn_pc, times)
    if coherent is not None:
        means = torch.cat((means[:Ns], means[Ns:] + coherent), 0)

    def pick(t, sl):
        return t if sl is all_sl else t[sl]  # a full slice would still cost a zeros+copy pair in backward

    def raster(sl, colors, bgs, mode):
        return _R.rasterization(means=pick(means, sl), quats=pick(quats, sl), scales=pick(scales, sl),
                                opacities=pick(opac, sl), colors=colors, backgrounds=bgs, viewmats=viewmat[None],
                                Ks=K[None], width=W, height=H, packed=False, render_mode=mode)

    rays = _rays_of(cam)

    def decode_ed(img, alphas):
        return decode(img, alphas, rays, w1, w2, True)  # views only: no select/zeros/copy in backward

    out = RenderResult({k: None for k in ("s_render", "s_depth", "d_render", "d_depth", "d_alpha", "d_means3d",
                                          "s_alpha", "blending_factor", "world_coordinates", "splat_center",
                                          "ori_flow", "ori_coord_map", "labels", "centroids")})

    ori_m2d = None
    if delta_exposure is not None and get_flow:
        o_means, o_quats, _, _, _ = _prep(stat_pc, dyn_pc, _times(cam, None, dev))
        _, ori_m2d, _, _, _ = _R.fully_fused_projection(means=o_means, covars=None, quats=o_quats, scales=scales,
                                                        viewmats=viewmat[None], Ks=K[None], width=W, height=H)

    # ONE projection and ONE tile binning / sort per render() call; the whole-set image comes from the single-set
    # compositor, the static-only / dynamic-only images (when asked for) from ONE layered walk over the same lists
    # (the reference: 5 rasterizations, :143-176, :201-214, :236-268)
    if sp is None:
        sp = _R.SharedProjection(means, quats, scales, opac, viewmat[None], K[None], W, H, pack_colors=cols,
                                 order=_enum_order(stat_pc, dyn_pc, means))
    # The intersection counts are still on their way to the host (speculative binning).  Compositing AND decoding are
    # enqueued before waiting for them, so that the device has the rest of the forward pass queued while the host
    # waits -- on small scenes (tens of thousands of splats) the step is host-bound and this wait was a bubble.
    # rows [0, Ns) are the static set: colour features cat(f_dc, 0.0 * f_t) (scene/gaussian_model.py:244-246) -- the
    # backward compositor leaves their three dead channels out (include/mobgs_hip.h MobgsTuning.static_rows)
    sp.static_rows = Ns
    rebuilds = sp.tl.rebuilds
    sp.tl.defer = True
    try:
        # (the decoder runs as the epilogue of the compositing kernel when the camera gives pinhole rays: one launch)
        img, alphas, rendered, depth = sp.composite_decode(cols, bg1, rays, w1, w2)
    finally:
        sp.tl.defer = False
    sp.tl.resolve()
    if sp.tl.rebuilds != rebuilds:  # arena too small (first frame / scene grew): lists were rebuilt, do it again
        img, alphas, rendered, depth = sp.composite_decode(cols, bg1, rays, w1, w2)
    info = sp.meta()
    radii = info["radii"].squeeze(0)
    _keep_grad(info["means2d"])
    out["render"] = rendered
    out["depth"] = depth.unsqueeze(0)
    if get_dynamic:
        out["d_means3d"] = means[dyn_sl]
    if get_static:
        out["s_depth"] = rendered[..., -1]  # reference quirk (:250): slices the decoded image -> [3,H]

    def alpha_pass(a):
        """The reference's ones-colour pass (:163-177, :255-269): sum_i w_i + T_final * bg = (1 - T) + T * bg."""
        return a + (1.0 - a) * bg[0]

    def aux_images(cols=cols):
        res = {}
        if fuse_prep and not (FUSE_LAYERS and _R.CLASS_PASSES):
            cols = _prep(stat_pc, dyn_pc, times)[4]   # (switched off since the render: the features as an array after all)
        if FUSE_LAYERS:
            imgs, alps = sp.composite_layers(cols, Ns, bg1, want_static=get_static, want_dynamic=get_dynamic)
            if get_dynamic:
                res["d_render"], d_depth = decode_ed(imgs[2], alps[2])
                res["d_depth"] = d_depth.unsqueeze(0)
                res["d_alpha"] = alpha_pass(alps[2])
            if get_static:
                res["s_render"], _ = decode_ed(imgs[1], alps[1])
                res["s_alpha"] = alpha_pass(alps[1])
            return res
        # call-for-call like the reference (A/B tests): one rasterization per set + a ones-colour alpha pass
        for name, sl, on in (("d", dyn_sl, get_dynamic), ("s", stat_sl, get_static)):
            if not on:
                continue
            x_img, x_a, _ = _raster_acc(raster, sl, cols[sl], bg1)
            res[name + "_render"], x_depth = decode_ed(x_img, x_a)
            if name == "d":
                res["d_depth"] = x_depth.unsqueeze(0)
            ones = torch.ones(cols[sl].shape[0], 1, device=dev)
            res[name + "_alpha"] = raster(sl, ones, bg[0:1][None], "RGB")[0][..., 0]
        return res

    aux_keys = (["d_render", "d_depth", "d_alpha"] if get_dynamic else []) + \
               (["s_render", "s_alpha"] if get_static else [])

    if ori_m2d is not None:
        flow_2d = (ori_m2d - info["means2d"].detach()).squeeze(0)
        flow_img, _ = _R.rasterize_to_pixels(sp.means2d, sp.conics, flow_2d, opac, sp.radii, sp.tl, W, H)  # same lists
        out["ori_flow"] = flow_img
        out["ori_coord_map"] = _pixel_grid(cam, W, H, flow_img) + flow_img

    out.update({"viewspace_points": info["means2d"], "radii": radii, "means_3d": means[dyn_sl]})
    if fuse_prep:  # the colour features as an ARRAY only exist when somebody asks for them (train.py does not)
        out.defer(["colors_precomp_final"], lambda: {"colors_precomp_final": _prep(stat_pc, dyn_pc, times)[4]})
    else:
        out["colors_precomp_final"] = cols
    if LAZY_AUX:  # train.py reads these two from the mid sub-frame only (:448-456), not from the 8 latent ones
        out.defer(["visibility_filter"], lambda: {"visibility_filter": radii > 0})
        out.defer(["means_3d_final"], lambda: {"means_3d_final": means * 1e2})
    else:
        out.update({"visibility_filter": radii > 0, "means_3d_final": means * 1e2})
    if aux_keys:
        if LAZY_AUX and FUSE_LAYERS:
            out.defer(aux_keys, aux_images)
        else:
            out.update(aux_images())
    return out


_bgK_cache = {}  # K -> DerivedCache of the [K,9] background rows


@_scoped
def render_many(viewpoint_cameras, stat_pc, dyn_pc, pipe, bg_color, delta_exposures=None):
    """[render(cam_k, stat_pc, dyn_pc, pipe, bg_color, delta_exposure=d_k) for k] in lean mode -- the K latent sub-frames
    of one blurry view (train.py:502-518: same Gaussians at K exposure times through K warped cameras) -- as ONE batch of
    C = K cameras: ONE per-splat prep for the K instants, ONE projection (every camera with its own positions / rotations / colours of
    the N splats: MobgsTuning.geometry_per_camera), ONE binning + sort, ONE compositing pass forward and backward over
    K x the tiles, K decodes.  Not in the reference (it renders the sub-frames one call at a time).  On large scenes a
    single render already fills the chip and nothing is gained; at the reference's own operating point (512x288, ~30 k
    splats: 576 tiles, every kernel at its launch floor) the sub-frames' renders stop being latency-bound.
    Images are bit-identical to separate render() calls; gradients agree to summation order.
    -> list of K dicts {"render" [3,H,W], "depth" [1,H,W], "radii" [N], "viewspace_points", "viewspace_index",
    "visibility_filter"}.  NOTE "viewspace_points" is the ONE [K,N,2] tensor of the batch, shared by the K dicts (its
    gradient arrives in one piece), NOT render()'s [1,N,2]: row "viewspace_index" belongs to the dict's camera.  Code
    written for render() -- `viewspace_point_tensor.grad.squeeze(0)[:Ns]`, train.py:637-646 -- must go through
    viewspace_grad(out), which returns that camera's [N,2] gradient."""
    cams = list(viewpoint_cameras)
    K = len(cams)
    deltas = list(delta_exposures) if delta_exposures is not None else [None] * K
    dev = _device_of(dyn_pc)
    W, H = int(cams[0].image_width), int(cams[0].image_height)
    if any((int(c.image_width), int(c.image_height)) != (W, H) for c in cams):
        raise ValueError("render_many: all cameras must share one image size")
    bg1 = _bg9(bg_color)
    bgK = _bgK_cache.setdefault(K, DerivedCache()).get((bg1,), lambda: bg1.expand(K, 9).contiguous())
    w1, w2 = _decoder_weights(dyn_pc)
    # the K instants in ONE prep launch (and one in the backward pass, which sums the leaf gradients in instant order)
    means, quats, scales, opac, cols = _prep(stat_pc, dyn_pc, torch.stack([_times(c, d, dev) for c, d in zip(cams, deltas)]))
    viewmats = torch.stack([c.world_view_transform.transpose(0, 1) for c in cams])
    Ks = torch.stack([c.K for c in cams])
    sp = _R.SharedProjection(means, quats, scales, opac, viewmats, Ks, W, H, pack_colors=cols,
                             order=_enum_order(stat_pc, dyn_pc, means, K))
    sp.static_rows = stat_pc.get_xyz.shape[0]

    def composite_and_decode():
        rays = _rays_of_many(cams) if K > 1 else None   # (one image: the single-image call, [3,H,W] out)
        if rays is not None:   # the K images decoded by the compositing launch itself (one decoder launch in backward)
            _, _, rgb, depth = sp.composite_decode(cols, bgK, rays, w1, w2)  # [K,3,H,W], [K,H,W]
            return list(zip(rgb.unbind(0), depth.unbind(0)))
        img, alphas = sp.composite(cols, bgK)                  # [K,H,W,10], [K,H,W,1]
        # unbind: ONE autograd node whose backward stacks the K cotangents (a select per k would zero-fill and copy
        # the whole batch K times)
        return [decode(i, a, _rays_of(c), w1, w2, True) for i, a, c in zip(img.unbind(0), alphas.unbind(0), cams)]

    rebuilds = sp.tl.rebuilds
    sp.tl.defer = True
    try:
        decoded = composite_and_decode()
    finally:
        sp.tl.defer = False
    sp.tl.resolve()
    if sp.tl.rebuilds != rebuilds:   # arena too small (first frame / scene grew): lists were rebuilt, do it again
        decoded = composite_and_decode()
    info = sp.meta()
    _keep_grad(info["means2d"])
    outs = []
    for k, (rendered, depth) in enumerate(decoded):
        radii = info["radii"][k]
        outs.append({"render": rendered, "depth": depth.unsqueeze(0), "radii": radii,
                     "viewspace_points": info["means2d"], "viewspace_index": k, "visibility_filter": radii > 0})
    return outs


def viewspace_grad(out):
    """[N,2] gradient of the 2-D means of ONE output dict -- of render() ("viewspace_points" [1,N,2]) or of render_many()
    (the batch's shared [K,N,2] tensor, row out["viewspace_index"]): what train.py:637-646 reads as
    `viewspace_point_tensor.grad.squeeze(0)`.  None before backward."""
    g = out["viewspace_points"].grad
    if g is None:
        return None
    return g[out["viewspace_index"]] if "viewspace_index" in out else g.squeeze(0)


def _raster_acc(raster, sl, colors, bgs):
    """'RGB+D' compositing (9 feature channels + accumulated depth); the ED division happens in the decoder."""
    img, alphas, info = raster(sl, colors, bgs, "RGB+D")
    return img, alphas.squeeze(-1), info


_pixel_grid_cache = {}


def _pixel_grid(cam, W, H, like):
    """cam.get_pixels(W, H, use_center=False) (/root/reference/scene/cameras.py:206-213: integer pixel corners,
    a function of (W, H) only) built once per size ON the device; the reference rebuilds it with numpy and copies
    11 MB host->device on every call."""
    key = (W, H, str(like.device), like.dtype)
    g = _pixel_grid_cache.get(key)
    if g is None:
        if len(_pixel_grid_cache) > 8:
            _pixel_grid_cache.clear()
        xs = torch.arange(W, dtype=torch.float32, device=like.device)
        ys = torch.arange(H, dtype=torch.float32, device=like.device)
        g = torch.stack([xs[None, :].expand(H, W), ys[:, None].expand(H, W)], dim=-1).to(like.dtype).contiguous()
        _pixel_grid_cache[key] = g
    return g


# get_flow()'s images feed ONE loss term, the flow-consistency loss, whose weight is 0 in the shipped configurations
# (/root/reference/train.py:675, arguments/stereo/seesaw.py:18 lambda_flow_loss = 0): the calls are made, their cotangents
# arrive as exact zeros.  The compositing nodes recorded by the functions below therefore probe their cotangents on the
# device in backward and skip the pass when all are zero (rendering.zero_cotangent_gate; bit-identical gradients
# otherwise, exact zeros when skipped; no host synchronisation).  MOBGS_ZERO_GATE=0 switches it off (A/B).
ZERO_GATE = __import__("os").environ.get("MOBGS_ZERO_GATE", "1") != "0"


# ... and, outside HIP-graph captures, the callers' side of the same question is answered ONCE per call (group): the
# outputs leave through _FlowHead, whose backward probes all their cotangents together and, when every one is zero,
# hands autograd `None` for all of them -- nothing upstream runs at all (no compositing, decoder, projection, prep or
# glue kernels; the leaves' .grad stay untouched, i.e. the term adds exactly nothing).  Costs one 4-byte read-back per
# call group in backward; MOBGS_FLOW_HOST_GATE=0 leaves only the device-side gate.
FLOW_HOST_GATE = __import__("os").environ.get("MOBGS_FLOW_HOST_GATE", "1") != "0"


class _FlowHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *tensors):
        ctx.set_materialize_grads(False)
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        if not FLOW_HOST_GATE:
            return grads
        live = [g for g in grads if g is not None]
        if not live or not all(g.is_cuda and g.dtype == torch.float32 for g in live):
            return grads
        if torch.cuda.is_current_stream_capturing():
            return grads
        if _R.cotangents_all_zero(live):
            return (None,) * len(grads)
        return grads


def _flow_head(outs):
    """outs: list of per-call output tuples -> the same structure behind one _FlowHead node."""
    if not (ZERO_GATE and FLOW_HOST_GATE and torch.is_grad_enabled()):
        return outs
    flat = [t for o in outs for t in o]
    if not any(t.requires_grad for t in flat):
        return outs
    flat = _FlowHead.apply(*flat)
    res, i = [], 0
    for o in outs:
        res.append(tuple(flat[i:i + len(o)]))
        i += len(o)
    return res


def _gated(fn):
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        with _R.zero_cotangent_gate(ZERO_GATE):
            return fn(*a, **k)
    return wrapped




@_gated
def _flow_mid_state(cam, stat_pc, dyn_pc, dev):
    """Projection + tile lists of the scene at the camera's own (mid-exposure) time: the part of get_flow() that does
    not depend on delta_exposure."""
    W, H = int(cam.image_width), int(cam.image_height)
    viewmat = cam.world_view_transform.transpose(0, 1)
    mid_m, mid_q, scales, opac, cols = _prep(stat_pc, dyn_pc, _times(cam, None, dev))
    sp = _R.SharedProjection(mid_m, mid_q, scales, opac, viewmat[None], cam.K[None], W, H,
                             order=_enum_order(stat_pc, dyn_pc, mid_m))
    sp.flow_cols = cols  # the colour features at the mid time (get_flow_many: the call with exposure offset 0)
    sp.static_rows = stat_pc.get_xyz.shape[0]
    return sp


# Implicit sharing of the mid-exposure state between SEPARATE get_flow() calls: train.py:570-579
# calls get_flow() nine times per view with nine exposure offsets; the prep / projection / tile lists at the camera's own
# time do not depend on the offset.  get_flow_many() shares them explicitly; the unchanged caller gets the same through a
# one-entry cache keyed on the camera, both Gaussian sets and the (storage, autograd version, length) of every tensor
# that enters the state -- an optimiser step, densification or a new camera pose changes the key.  An entry whose graph
# has been back-propagated through is dropped (its buffers are gone).  Not visible to the key: writes through `.data`
# (they bypass the version counter) -- call invalidate_flow_cache() after such a write, or set MOBGS_FLOW_MID_CACHE=0.
FLOW_MID_CACHE = __import__("os").environ.get("MOBGS_FLOW_MID_CACHE", "1") != "0"
_mid_cache = {}
mid_cache_stats = {"hits": 0, "misses": 0}


_param_epoch = [0]


def invalidate_flow_cache():
    _mid_cache.clear()


def parameters_changed():
    """Called by everything in this package that writes Gaussian parameters WITHOUT moving Tensor._version -- the fused
    Adam kernel (optim.fused_adam_step), in-place row permutations / densification surgery (GaussianParams.spatial_sort_,
    densify.TrainableGaussians._rebuild) -- so that state cached across calls (the shared mid-exposure projection of
    get_flow) is never reused over new values.  Foreign `.data` writes: call it (or invalidate_flow_cache())
    yourself, or set MOBGS_FLOW_MID_CACHE=0."""
    _param_epoch[0] += 1
    _mid_cache.clear()   # (also releases the projection graph and arenas the entry pins)


def _sig(t):
    return (t.data_ptr(), t._version, tuple(t.shape), t.requires_grad) if torch.is_tensor(t) else t


def _mid_signature(cam, stat_pc, dyn_pc):
    ts = (stat_pc._xyz, stat_pc._scaling, stat_pc._rotation, stat_pc._opacity, stat_pc._features_dc, stat_pc._features_t,
          dyn_pc.get_control_xyz, dyn_pc.current_control_num, dyn_pc._scaling, dyn_pc._rotation, dyn_pc._omega,
          dyn_pc._opacity, dyn_pc._features_dc, dyn_pc._features_t, dyn_pc.get_trbfcenter, cam.world_view_transform,
          cam.K, getattr(cam, "static_times", None))
    return tuple(_sig(t) for t in ts) + (float(cam.time), float(cam.max_time), int(cam.image_width),
                                         int(cam.image_height), torch.is_grad_enabled(), _param_epoch[0])


def _shared_mid_state(cam, stat_pc, dyn_pc, dev):
    if not FLOW_MID_CACHE or torch.cuda.is_current_stream_capturing():
        return _flow_mid_state(cam, stat_pc, dyn_pc, dev)
    sig = _mid_signature(cam, stat_pc, dyn_pc) + (_R.stream_int(),)
    e = _mid_cache.get("entry")
    if e is not None and e["sig"] == sig and e["cam"]() is cam and e["stat"]() is stat_pc and e["dyn"]() is dyn_pc \
            and not e["used"][0]:
        mid_cache_stats["hits"] += 1
        return e["sp"]
    mid_cache_stats["misses"] += 1
    sp = _flow_mid_state(cam, stat_pc, dyn_pc, dev)
    used = [False]
    if sp.means2d.requires_grad:
        def _mark(g, used=used):
            used[0] = True   # a backward pass reached the shared state: its saved buffers are being released
            return g
        sp.means2d.register_hook(_mark)
    _mid_cache["entry"] = {"sig": sig, "cam": weakref.ref(cam), "stat": weakref.ref(stat_pc), "dyn": weakref.ref(dyn_pc),
                           "sp": sp, "used": used}
    return sp


@_scoped
@_gated
def get_flow(viewpoint_camera, stat_pc, dyn_pc, pipe, bg_color, delta_exposure=None, _mid=None, _defer_mid=False):
    """/root/reference/gaussian_renderer/__init__.py:318-492 ->
    (exp2mid_coord_map [1,H,W,2], mid2exp_coord_map [1,H,W,2], latent_img [3,H,W], latent_alpha [1,H,W]).
    While gradients are recorded (and the zero-cotangent host gate is on, the default) the four maps of a call are outputs of
    ONE autograd node that returns views (_FlowHead): use them out of place.  /root/reference/train.py:570-579, :658-668
    does -- it concatenates the nine calls' maps (torch.cat: a fresh tensor) before normalising `coord[..., 0] /= W - 1` in
    place; an in-place edit of a RETURNED map itself raises PyTorch's "output of a function that returns multiple views"
    error (tests/test_host_logic_cpu.py pins both).  MOBGS_FLOW_HOST_GATE=0 returns ordinary tensors."""
    cam = viewpoint_camera
    dev = _device_of(dyn_pc)
    W, H = int(cam.image_width), int(cam.image_height)
    viewmat = cam.world_view_transform.transpose(0, 1)
    K = cam.K
    bg1 = _bg9(bg_color)
    bg = bg1[0]
    w1, w2 = _decoder_weights(dyn_pc)
    Ns = stat_pc.get_xyz.shape[0]
    exp_m, exp_q, scales, opac, exp_c = _prep(stat_pc, dyn_pc, _times(cam, delta_exposure, dev))

    # two projections + two tile binnings of the whole set (the reference: 2 explicit projections + 4 rasterizations)
    sp_exp = _R.SharedProjection(exp_m, exp_q, scales, opac, viewmat[None], K[None], W, H,
                                 order=_enum_order(stat_pc, dyn_pc, exp_m))
    # static rows in the 12-channel pass below: f_t channels 0.0 * f_t, flow channels x_mid - x_exp = 0 exactly (a static
    # splat projects alike at both exposures; the +g / -g its flow gradient sends through the two identical projections
    # cancel) -- five dead channels (MobgsTuning.static_rows; verified per entry by the kernel)
    sp_exp.static_rows = Ns
    sp_mid = _mid if _mid is not None else _shared_mid_state(cam, stat_pc, dyn_pc, dev)

    def splat(sp, colors):
        return _R.rasterize_to_pixels(sp.means2d, sp.conics, colors, opac, sp.radii, sp.tl, W, H)[0]

    # dynamic-only coverage (:477-490: a rasterization of the dynamic splats alone with a ones colour): a
    # class-restricted walk over the exposure-time lists of the whole set -- the same image without projecting,
    # binning and sorting the dynamic third a second time; sum_i w_i + T_final * bg = (1 - T) + T * bg
    latent_alpha = sp_exp.class_alpha(Ns, 2, background=bg1[:, :1])
    e2m = (sp_mid.means2d - sp_exp.means2d).squeeze(0)
    # the exposure-time lists are walked ONCE for the 9 colour features and the 2 flow channels (the reference: one
    # rasterization each, :436-452 and :461-476; channels accumulate independently, so the images are identical)
    img12, alphas = sp_exp.composite(torch.cat([exp_c, e2m], dim=-1), _bg11(bg1))  # [1,H,W, 9 + 2 + depth]
    # the decoder reads the 9 feature channels; the 2 flow channels leave through the same autograd node
    latent_img, e2m_img = decode_with_channels(img12, alphas, _rays_of(cam), w1, w2, 9, 2)
    pix = _pixel_grid(cam, W, H, e2m_img)
    exp2mid = pix + e2m_img[None]
    # get_flow_many splats the mid-exposure flows of several calls in one walk over the shared lists
    if _defer_mid:
        return exp2mid, (pix, e2m), latent_img, latent_alpha
    return _flow_head([(exp2mid, pix + splat(sp_mid, -e2m), latent_img, latent_alpha)])[0]


@_scoped
@_gated
def get_flow_many(viewpoint_camera, stat_pc, dyn_pc, pipe, bg_color, delta_exposures):
    """[get_flow(..., delta_exposure=d) for d in delta_exposures] -- the 9 calls per view of train.py:570-579 -- with
    the mid-exposure prep / projection / tile lists (which do not depend on d) built once and shared by all of them,
    and the mid-to-exposure flow maps of up to 8 calls splatted in ONE walk over those lists (2 channels each: the
    cost of a compositing pass is its alpha evaluations, not its channels -- nine 2-channel passes take 9 x 0.58 ms
    forward + backward, one 16-channel pass ~1.5 ms).  Not in the reference; the images are identical (channels
    accumulate independently), gradients equal up to summation order."""
    cam = viewpoint_camera
    mid = _flow_mid_state(cam, stat_pc, dyn_pc, _device_of(dyn_pc))
    W, H = int(cam.image_width), int(cam.image_height)
    outs = [None] * len(delta_exposures)
    off_mid = []
    for i, d in enumerate(delta_exposures):
        if not torch.is_tensor(d) and float(d) == 0.0:
            # the mid exposure itself (train.py's fifth call): the exposure-time state IS the mid state, both flows
            # are identically zero (their gradients cancel term by term), so the call needs no projection, binning,
            # flow channels or flow splat of its own -- one 10-channel walk over the shared lists
            outs[i] = _flow_at_mid(cam, mid, stat_pc.get_xyz.shape[0], dyn_pc, bg_color, W, H)
        else:
            off_mid.append(i)
    if BATCH_FLOW_EXPOSURES and len(off_mid) > 1:
        # round 3: the exposure-time halves of all those calls as ONE batch of G "cameras" (the same pose G times,
        # per-camera geometry: the splats at G exposure times) -- one projection, binning, sort, 12-channel compositing
        # pass and coverage pass for all of them instead of G (see render_many)
        for i, o in zip(off_mid, _get_flow_exposures(cam, stat_pc, dyn_pc, bg_color, [delta_exposures[i] for i in off_mid],
                                                     mid)):
            outs[i] = o
    else:
        for i in off_mid:
            outs[i] = list(get_flow(cam, stat_pc, dyn_pc, pipe, bg_color, delta_exposure=delta_exposures[i], _mid=mid,
                                    _defer_mid=True))
    pending = [o for o in outs if isinstance(o[1], tuple)]
    for g0 in range(0, len(pending), _FLOW_GROUP):
        grp = pending[g0:g0 + _FLOW_GROUP]
        cols = -torch.cat([o[1][1] for o in grp], dim=-1)  # [N, 2 * len(grp)]  (one negation, not one per call)
        img = _R.rasterize_to_pixels(mid.means2d, mid.conics, cols, mid.opacities, mid.radii, mid.tl, W, H)[0]
        # split, not slices: its backward is ONE concatenation of the 2-channel cotangents instead of a zero image, a
        # strided copy and an add per call
        # one broadcast add of the pixel grid for the whole group (was one add per call); the maps are then views
        tot = img.unflatten(-1, (len(grp), 2)) + grp[0][1][0][:, :, None, :]
        for o, part in zip(grp, tot.unbind(-2)):
            o[1] = part
    return _flow_head([tuple(o) for o in outs])


BATCH_FLOW_EXPOSURES = True


def _get_flow_exposures(cam, stat_pc, dyn_pc, bg_color, deltas, mid):
    """[get_flow(cam, ..., delta_exposure=d, _mid=mid, _defer_mid=True) for d in deltas] with the exposure-time work of
    all G calls batched: -> list of [exp2mid, (pix, e2m_g), latent_img, latent_alpha]."""
    dev = _device_of(dyn_pc)
    W, H = int(cam.image_width), int(cam.image_height)
    G = len(deltas)
    viewmat = cam.world_view_transform.transpose(0, 1)
    bg1 = _bg9(bg_color)
    w1, w2 = _decoder_weights(dyn_pc)
    Ns = stat_pc.get_xyz.shape[0]
    means, quats, scales, opac, cols = _prep(stat_pc, dyn_pc, torch.stack([_times(cam, d, dev) for d in deltas]))  # cols [G,N,9]
    sp = _R.SharedProjection(means, quats, scales, opac, viewmat[None].expand(G, 4, 4), cam.K[None].expand(G, 3, 3), W, H,
                             order=_enum_order(stat_pc, dyn_pc, means, G))
    sp.static_rows = Ns   # (see get_flow)
    bgG = _bgK_cache.setdefault(G, DerivedCache()).get((bg1,), lambda: bg1.expand(G, 9).contiguous())
    bg11 = _bgK_cache.setdefault(("11", G), DerivedCache()).get(
        (bg1,), lambda: torch.cat([bg1, bg1.new_zeros(1, 2)], dim=-1).expand(G, 11).contiguous())
    latent_alpha = sp.class_alpha(Ns, 2, background=bgG[:, :1])    # [G,H,W]
    e2m = mid.means2d - sp.means2d                                 # [G,N,2]
    img12, alphas = sp.composite(torch.cat([cols, e2m], dim=-1), bg11)   # [G,H,W,12]
    rays = _rays_of(cam)
    outs = []
    # ONE decoder launch for the G images (same camera: shared rays); unbind, not [g]: one autograd node whose backward
    # stacks the G cotangents (a select per g zero-fills and copies the whole array G times)
    latent_imgs, e2m_imgs = decode_with_channels(img12, alphas, rays, w1, w2, 9, 2)    # [G,3,H,W], [G,H,W,2]
    pix = _pixel_grid(cam, W, H, e2m_imgs)
    exp2mid = pix + e2m_imgs                                                             # one add for the G maps
    for x2m, li, la, e2m_g in zip(exp2mid.unbind(0), latent_imgs.unbind(0), latent_alpha.unbind(0), e2m.unbind(0)):
        outs.append([x2m[None], (pix, e2m_g), li, la[None]])
    return outs


def _flow_at_mid(cam, mid, Ns, dyn_pc, bg_color, W, H):
    """get_flow(delta_exposure = 0) from the shared mid-exposure state: [exp2mid, mid2exp, latent_img, latent_alpha]."""
    bg1 = _bg9(bg_color)
    w1, w2 = _decoder_weights(dyn_pc)
    latent_alpha = mid.class_alpha(Ns, 2, background=bg1[:, :1])
    img10, alphas = mid.composite(mid.flow_cols, bg1)
    pix = _pixel_grid(cam, W, H, img10)
    latent_img, _ = decode(img10, alphas, _rays_of(cam), w1, w2, False)
    grid = pix.unsqueeze(0)  # [1,H,W,2] like pix + image
    # part of the autograd graph like the other calls' maps (a caller may hand them to autograd.backward directly);
    # their gradient is identically zero
    return [_ZeroGradCopy.apply(grid, mid.means2d), _ZeroGradCopy.apply(grid, mid.means2d), latent_img, latent_alpha]


class _ZeroGradCopy(torch.autograd.Function):
    """A copy of `value` that belongs to `anchor`'s autograd graph and passes no gradient on."""

    @staticmethod
    def forward(ctx, value, anchor):
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        return None, None


_FLOW_GROUP = 8  # exposures per mid-list walk: 16 channels, the widest compositor build below the 26-channel one


def get_flow_static(source_camera, target_camera, splat_camera, stat_pc, dyn_pc, pipe, bg_color):
    """/root/reference/gaussian_renderer/__init__.py:494-552 -> (flow_2d [Ns,2], rendered_flow [1,H,W,2])."""
    means, scales = stat_pc.get_xyz, stat_pc.get_scaling
    quats, opac = stat_pc._rotation, stat_pc.get_opacity.squeeze(-1)
    K = source_camera.K

    def project(cam):
        return _R.fully_fused_projection(means=means, covars=None, quats=quats, scales=scales,
                                         viewmats=cam.world_view_transform.transpose(0, 1)[None], Ks=K[None],
                                         width=int(cam.image_width), height=int(cam.image_height))[1]

    flow_2d = (project(source_camera) - project(target_camera)).squeeze(0)
    img = _R.rasterization(means=means, quats=quats, scales=scales, opacities=opac, colors=flow_2d,
                           backgrounds=None, viewmats=splat_camera.world_view_transform.transpose(0, 1)[None],
                           Ks=K[None], width=int(splat_camera.image_width), height=int(splat_camera.image_height),
                           packed=False, render_mode="RGB")[0]
    return flow_2d, img

get_flow_static = _scoped(get_flow_static, stat_at=3)

