"""Densification and optimiser surgery of a Gaussian set.

`TrainableGaussians` carries the 14 per-splat optimiser groups of the reference's GaussianModel
(/root/reference/scene/gaussian_model.py:598-617), a torch.optim.Adam over them (eps 1e-15, :645) and the five
per-splat statistics arrays, and offers the reference's method names with the reference's semantics:

    training_setup(opt)                      :590-661 (per-splat groups + decoder)
    add_densification_stats(vsp_grad, vis)   :1352-1356   (+ update_max_radii(): helper_train.py:263)
    densify_and_clone(grads, thr, extent)    :1480-1506
    densify_and_splitv2(grads, thr, extent, N)   :1207-1244
    densify_pruneclone(thr, min_opacity, extent, max_screen_size, splitN)   :1417-1434
    prune_points(mask)                       :1066-1089
    reset_opacity()                          :897-903

Where the reference re-allocates every group with ~50 index / cat / repeat calls per operation, every field lives
here in a capacity-sized buffer pair; a resize builds ONE row list and moves all fields (parameters, Adam moments,
statistics) with one mobgs_rows_gather launch into the other buffer of the pair.  Parameters are views [:n] of the
current buffers; as in the reference they are NEW nn.Parameter objects after each resize and the optimiser's
groups/state are re-keyed to them.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional

import torch
from torch import nn

from . import _lib
from ._lib import check, ptr, stream
from .gaussian_model import GaussianParams

# optimiser group name -> attribute (reference :598-617); shapes per splat
GROUPS = [("xyz", "_xyz"), ("control_xyz", "control_xyz"), ("current_control_num", "current_control_num"),
          ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("f_t", "_features_t"), ("opacity", "_opacity"),
          ("scaling", "_scaling"), ("rotation", "_rotation"), ("omega", "_omega"), ("zeta", "_zeta"),
          ("trbf_center", "_trbf_center"), ("trbf_scale", "_trbf_scale"), ("motion", "_motion")]
STATS = ["xyz_gradient_accum", "denom", "max_radii2D", "_deformation_accum"]  # reset by every append
_LR = {"xyz": lambda o, s: o.position_lr_init * s, "control_xyz": lambda o, s: 10 * o.position_lr_init * s,
       "current_control_num": lambda o, s: 0.0, "f_dc": lambda o, s: o.feature_lr,
       "f_rest": lambda o, s: o.feature_lr / 20.0, "f_t": lambda o, s: o.featuret_lr,
       "opacity": lambda o, s: o.opacity_lr, "scaling": lambda o, s: o.scaling_lr,
       "rotation": lambda o, s: o.rotation_lr, "omega": lambda o, s: o.omega_lr, "zeta": lambda o, s: o.zeta_lr,
       "trbf_center": lambda o, s: o.trbfc_lr, "trbf_scale": lambda o, s: o.trbfs_lr,
       "motion": lambda o, s: o.position_lr_init * s * 0.5 * o.movelr}


class _Field:
    """One per-splat array: two capacity-sized buffers (current / spare) and the per-row shape."""

    def __init__(self, data: torch.Tensor, capacity: int, zero_new: bool):
        self.row_shape = tuple(data.shape[1:])
        self.dtype = data.dtype
        self.zero_new = zero_new
        self.buf = [torch.empty((capacity,) + self.row_shape, dtype=data.dtype, device=data.device) for _ in range(2)]
        self.cur = 0
        self.buf[0][:data.shape[0]] = data

    @property
    def row_bytes(self) -> int:
        n = 1
        for d in self.row_shape:
            n *= d
        return n * self.buf[0].element_size()

    def view(self, n: int) -> torch.Tensor:
        return self.buf[self.cur][:n]

    def grow(self, capacity: int, n: int):
        for k in range(2):
            old = self.buf[k]
            new = torch.empty((capacity,) + self.row_shape, dtype=self.dtype, device=old.device)
            if k == self.cur:
                new[:n] = old[:n]
            self.buf[k] = new


class TrainableGaussians(GaussianParams):
    def __init__(self, params: Dict[str, torch.Tensor], dynamic: Optional[Dict[str, torch.Tensor]] = None,
                 decoder=None, device="cuda", capacity_factor: float = 1.5):
        super().__init__(params, dynamic, decoder, device=device, requires_grad=False)
        n = self._xyz.shape[0]
        dev = self._xyz.device
        z = lambda *s: torch.zeros(n, *s, dtype=torch.float32, device=dev)  # noqa: E731
        dyn = dynamic or {}
        self._opacity = self._opacity.reshape(n, 1)
        self._features_rest = dyn["f_rest"].to(dev) if "f_rest" in dyn else z(0, 3)
        self._zeta = dyn["zeta"].to(dev) if "zeta" in dyn else z(1)
        self._trbf_scale = dyn["trbf_scale"].to(dev) if "trbf_scale" in dyn else z(1)
        self._motion = dyn["motion"].to(dev) if "motion" in dyn else z(9)
        self.current_control_num = self.current_control_num.reshape(n, 1)
        self._deformation_table = (dyn["_deformation_table"].to(dev) if "_deformation_table" in dyn
                                   else torch.ones(n, dtype=torch.bool, device=dev))
        self.xyz_gradient_accum, self.denom, self._deformation_accum = z(1), z(1), z(3)
        self.max_radii2D = torch.zeros(n, dtype=torch.float32, device=dev)
        self.percent_dense = 0.01
        self.spatial_lr_scale = 1.0
        self.optimizer = None
        self._n = n
        self._capacity = max(int(n * capacity_factor) + 1024, n)
        self._factor = capacity_factor
        self._fields: Dict[str, _Field] = {}
        for g, attr in GROUPS:
            self._fields[g] = _Field(getattr(self, attr).detach(), self._capacity, False)
        self._fields["_deformation_table"] = _Field(self._deformation_table.to(torch.uint8), self._capacity, False)
        for s in STATS:
            self._fields[s] = _Field(getattr(self, s), self._capacity, True)
        self._publish(rekey=False)

    # ---- views <-> attributes ------------------------------------------------------------------------------------
    def _publish(self, rekey: bool):
        """Point the reference-named attributes at the current buffers; with an optimiser, re-key its groups and
        state to the new Parameters (what replace/cat/prune do group by group in the reference)."""
        n = self._n
        groups = {gr["name"]: gr for gr in self.optimizer.param_groups} if self.optimizer is not None else {}
        for g, attr in GROUPS:
            t = self._fields[g].view(n)
            if g == "current_control_num":
                new = t  # integer knot count: a plain tensor in the reference's optimiser group too (:601)
            else:
                new = nn.Parameter(t.requires_grad_(True))
            if g in groups:
                old = groups[g]["params"][0]
                st = self.optimizer.state.pop(old, None)
                if st is not None and "exp_avg" in st:
                    st["exp_avg"] = self._fields[g + ".exp_avg"].view(n)
                    st["exp_avg_sq"] = self._fields[g + ".exp_avg_sq"].view(n)
                    self.optimizer.state[new] = st
                groups[g]["params"][0] = new
            setattr(self, attr, new)
        self._deformation_table = self._fields["_deformation_table"].view(n).view(torch.bool)
        for s in STATS:
            setattr(self, s, self._fields[s].view(n))

    def training_setup(self, training_args):
        """Per-splat groups + decoder of the reference's training_setup (:598-617, :645); the deformation / pose
        groups belong to modules outside this class and can be added with optimizer.add_param_group()."""
        self.percent_dense = training_args.percent_dense
        for s in ("xyz_gradient_accum", "denom", "_deformation_accum"):
            getattr(self, s).zero_()
        scale = self.spatial_lr_scale
        groups = [{"params": [getattr(self, attr)], "lr": _LR[g](training_args, scale), "name": g}
                  for g, attr in GROUPS]
        groups.append({"params": list(self.rgbdecoder.parameters()), "lr": training_args.rgb_lr, "name": "decoder"})
        from .optim import FusedAdam   # torch.optim.Adam whose step() is ONE launch (the caller's loop stays unchanged)
        self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)

    def adopt_optimizer_state(self):
        """Move the Adam moments torch allocated on the first step() into table fields (so that they are resized with
        the parameters).  Called lazily by every resize."""
        n = self._n
        for gr in self.optimizer.param_groups:
            g = gr["name"]
            if g not in self._fields or len(gr["params"]) != 1:
                continue
            st = self.optimizer.state.get(gr["params"][0], None)
            if st is None or "exp_avg" not in st:
                continue
            for k in ("exp_avg", "exp_avg_sq"):
                name = f"{g}.{k}"
                f = self._fields.get(name)
                if f is None:
                    f = self._fields[name] = _Field(st[k], self._capacity, True)
                elif st[k].data_ptr() != f.view(n).data_ptr():
                    f.view(n).copy_(st[k])
                st[k] = f.view(n)

    # ---- per-step statistics ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def add_densification_stats(self, viewspace_point_tensor, update_filter, radii=None):
        """gaussian_model.py:1352-1356; with `radii` also helper_train.py:263 (max_radii2D) in the same launch.
        `viewspace_point_tensor` is the GRADIENT tensor [N, >=2], as passed by train.py:814-817."""
        g = viewspace_point_tensor
        if g.dim() == 3:
            g = g[0]
        g = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()
        vis = update_filter.reshape(-1).to(torch.uint8) if update_filter is not None else None
        r = radii.reshape(-1).to(torch.int32).contiguous() if radii is not None else None
        check(_lib.load().mobgs_densify_stats(self._n, ptr(g), g.shape[-1], ptr(vis), ptr(r),
                                             ptr(self.xyz_gradient_accum), ptr(self.denom),
                                             ptr(self.max_radii2D) if r is not None else None, stream()),
              "mobgs_densify_stats")

    # ---- resize machinery -----------------------------------------------------------------------------------------
    def _indices_of(self, mask_u8: torch.Tensor, want: int):
        n = mask_u8.shape[0]
        idx = torch.empty(n, dtype=torch.int32, device=mask_u8.device)
        cnt = torch.empty(1, dtype=torch.int32, device=mask_u8.device)
        check(_lib.load().mobgs_mask_indices(n, ptr(mask_u8), want, ptr(idx), ptr(cnt), stream()),
              "mobgs_mask_indices")
        return idx[:int(cnt.item())]

    @torch.no_grad()
    def spatial_sort_(self) -> torch.Tensor:
        """GaussianParams.spatial_sort_ for the whole table: parameters, Adam moments and densification statistics move
        together in ONE gather.  `keep_sorted = True` makes every densification end with it."""
        from .rendering import spatial_order
        order = spatial_order(self.sort_positions()).to(torch.int32)
        self._rebuild(order, reset_stats=False)
        self.rows_coherent = self._n
        return order.long()

    keep_sorted = False

    def _rebuild(self, index: torch.Tensor, reset_stats: bool):
        """Table := rows `index` (int32; negative = NEW copy of row -(i+1)) of the current table."""
        self.rows_coherent = -1
        from . import gaussian_renderer as _GR
        _GR.parameters_changed()   # (rows move in place: no version counter sees it -- ADVICE r5)
        if self.optimizer is not None:
            self.adopt_optimizer_state()
        n_out = int(index.shape[0])
        if n_out > self._capacity:
            self._capacity = int(n_out * self._factor) + 1024
            for f in self._fields.values():
                f.grow(self._capacity, self._n)
        names = [k for k in self._fields if not (reset_stats and k in STATS)]
        fields = [self._fields[k] for k in names]
        nf = len(fields)
        src = (ctypes.c_void_p * nf)(*[f.buf[f.cur].data_ptr() for f in fields])
        dst = (ctypes.c_void_p * nf)(*[f.buf[1 - f.cur].data_ptr() for f in fields])
        rb = (ctypes.c_int32 * nf)(*[f.row_bytes for f in fields])
        zn = (ctypes.c_int32 * nf)(*[1 if f.zero_new else 0 for f in fields])
        index = index.contiguous()
        check(_lib.load().mobgs_rows_gather(nf, src, dst, rb, zn, ptr(index) if n_out else None, n_out, 0, stream()),
              "mobgs_rows_gather")
        for f in fields:
            f.cur = 1 - f.cur
        if reset_stats:
            for s in STATS:
                self._fields[s].view(n_out).zero_()
        self._n = n_out
        self._publish(rekey=True)

    def _select(self, n_grads: int, grad_threshold: float, scene_extent: float):
        n = self._n
        dev = self._xyz.device
        clone = torch.empty(n, dtype=torch.uint8, device=dev)
        split = torch.empty(n, dtype=torch.uint8, device=dev)
        check(_lib.load().mobgs_densify_select(n, n_grads, ptr(self._sel_accum), ptr(self._sel_denom),
                                              ptr(self._fields["scaling"].view(n)), float(grad_threshold),
                                              float(self.percent_dense * scene_extent), ptr(clone), ptr(split),
                                              stream()), "mobgs_densify_select")
        return clone, split

    def _grads_as_ratio(self, grads):
        """The selection kernel forms accum / denom itself; explicit `grads` [n,1] are passed as grads / 1."""
        g = grads.reshape(-1).to(torch.float32).contiguous()
        self._sel_accum, self._sel_denom = g, torch.ones_like(g)
        return g.shape[0]

    # ---- the reference's operations ----------------------------------------------------------------------------
    @torch.no_grad()
    def prune_points(self, mask):
        keep = self._indices_of(mask.reshape(-1).to(torch.uint8), 0)
        coherent = self.rows_coherent == self._n
        self._rebuild(keep, reset_stats=False)
        if coherent:  # a subsequence of ordered rows is ordered
            self.rows_coherent = self._n

    @torch.no_grad()
    def densify_and_clone(self, grads, grad_threshold, scene_extent, *_unused, **_unused_kw):
        n_grads = self._grads_as_ratio(grads)
        clone, _ = self._select(n_grads, grad_threshold, scene_extent)
        sel = self._indices_of(clone, 1)
        every = torch.arange(self._n, dtype=torch.int32, device=sel.device)
        self._rebuild(torch.cat([every, -(sel + 1)]), reset_stats=True)
        if self.keep_sorted:
            self.spatial_sort_()

    @torch.no_grad()
    def densify_and_splitv2(self, grads, grad_threshold, scene_extent, N=2, samples=None):
        n_grads = self._grads_as_ratio(grads)
        _, split = self._select(n_grads, grad_threshold, scene_extent)
        keep = self._indices_of(split, 0)
        sel = self._indices_of(split, 1)
        self._split_into(keep, [], sel, N, samples)

    def _split_into(self, keep, clones, sel, N, samples):
        parents = -(sel + 1)
        index = torch.cat([keep, *clones, *([parents] * N)])
        first = int(keep.shape[0]) + sum(int(c.shape[0]) for c in clones)
        n_children = N * int(sel.shape[0])
        if samples is None and n_children:
            stds = torch.exp(self._fields["scaling"].view(self._n)[sel.long()]).repeat(N, 1)
            samples = torch.normal(mean=torch.zeros_like(stds), std=stds)
        self._rebuild(index, reset_stats=True)
        if n_children:
            samples = samples.to(torch.float32).contiguous()
            check(_lib.load().mobgs_split_children(n_children, first, N, ptr(samples),
                                                  ptr(self._fields["rotation"].view(self._n)),
                                                  ptr(self._fields["xyz"].view(self._n)),
                                                  ptr(self._fields["scaling"].view(self._n)), stream()),
                  "mobgs_split_children")
        if self.keep_sorted:
            self.spatial_sort_()

    @torch.no_grad()
    def densify_pruneclone(self, max_grad, min_opacity, extent, max_screen_size, splitN=2, samples=None):
        """:1417-1434: clone + splitv2 on grads = xyz_gradient_accum / denom (the reference also forms a prune mask
        there and never applies it).  One selection pass and ONE gather: the result equals the two separate steps
        ([kept originals | clones | split children])."""
        self._sel_accum, self._sel_denom = self.xyz_gradient_accum.reshape(-1), self.denom.reshape(-1)
        clone, split = self._select(self._n, max_grad, extent)
        keep = self._indices_of(split, 0)
        sel = self._indices_of(split, 1)
        cl = self._indices_of(clone, 1)
        self._split_into(keep, [-(cl + 1)], sel, splitN, samples)

    @torch.no_grad()
    def reset_opacity(self):
        op = torch.sigmoid(self._opacity)
        x = torch.min(op, torch.ones_like(op) * 0.01)
        new = torch.log(x / (1 - x))
        if torch.isnan(new).any():
            raise FloatingPointError("opacities_new is nan")  # the reference prints and exit()s here
        self._fields["opacity"].view(self._n).copy_(new)
        if self.optimizer is not None:
            self.adopt_optimizer_state()
            for k in ("opacity.exp_avg", "opacity.exp_avg_sq"):
                if k in self._fields:
                    self._fields[k].view(self._n).zero_()
        self._publish(rekey=True)

    # ---- checkpoint interchange (scene/gaussian_model.py:761-804, :934-1040) ---------------------------------------
    def save_ply(self, path: str):
        from . import ply_io
        ply_io.save_ply(self, path)

    @classmethod
    def from_ply(cls, path: str, device="cuda", decoder=None):
        from . import ply_io
        params, dynamic = ply_io.load_ply(path)
        pc = cls(params, dynamic, decoder=decoder, device=device)
        pt = path.replace(".ply", ".pt")
        if decoder is None and os.path.exists(pt):
            pc.rgbdecoder.load_state_dict(torch.load(pt, map_location=device))
        return pc

    # ---- introspection for tests ------------------------------------------------------------------------------------
    def table_state(self) -> Dict[str, torch.Tensor]:
        if self.optimizer is not None:
            self.adopt_optimizer_state()
        out = {k: f.view(self._n) for k, f in self._fields.items()}
        out["_deformation_table"] = out["_deformation_table"].view(torch.bool)
        return out
