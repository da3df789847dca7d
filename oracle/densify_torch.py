"""CPU restatement of the reference's densification / optimiser surgery (TEST INFRASTRUCTURE: imported only by
tests/ -- the product path is mobgs_amd/densify.py + csrc/densify.hip).

Pinned against tests/golden/densify.npz, which holds the states the reference's own GaussianModel produced for the
same inputs (tests/golden/make_golden.py:gen_densify, run in the build container).

State layout (plain tensors, first dimension = splat):
    state["params"][g]      the per-splat optimiser groups of scene/gaussian_model.py:598-617
    state["exp_avg"][g], state["exp_avg_sq"][g]   their Adam moments (groups without Adam state are absent)
    state["aux"][a]         xyz_gradient_accum [N,1], denom [N,1], max_radii2D [N], _deformation_table [N] bool,
                            _deformation_accum [N,3]
"""
from __future__ import annotations

import torch

GROUPS = ["xyz", "control_xyz", "current_control_num", "f_dc", "f_rest", "f_t", "opacity", "scaling", "rotation",
          "omega", "zeta", "trbf_center", "trbf_scale", "motion"]
AUX = ["xyz_gradient_accum", "denom", "max_radii2D", "_deformation_table", "_deformation_accum"]


def add_densification_stats(state, viewspace_grad, visible, radii):
    """helper_train.py:263-264 + scene/gaussian_model.py:1352-1356."""
    a = state["aux"]
    a["max_radii2D"][visible] = torch.max(a["max_radii2D"][visible], radii[visible])
    a["xyz_gradient_accum"][visible] += torch.norm(viewspace_grad[visible, :2], dim=-1, keepdim=True)
    a["denom"][visible] += 1


def mean_grads(state):
    """scene/gaussian_model.py:1418-1419."""
    g = state["aux"]["xyz_gradient_accum"] / state["aux"]["denom"]
    g[g.isnan()] = 0.0
    return g


def _append(state, new, new_table):
    """densification_postfix + cat_tensors_to_optimizer (scene/gaussian_model.py:1091-1155): rows appended, their
    Adam moments zero, the per-splat statistics of ALL rows reset, the deformation table extended."""
    for g in GROUPS:
        state["params"][g] = torch.cat((state["params"][g], new[g]), 0)
        if g in state["exp_avg"]:
            state["exp_avg"][g] = torch.cat((state["exp_avg"][g], torch.zeros_like(new[g])), 0)
            state["exp_avg_sq"][g] = torch.cat((state["exp_avg_sq"][g], torch.zeros_like(new[g])), 0)
    n = state["params"]["xyz"].shape[0]
    a = state["aux"]
    a["_deformation_table"] = torch.cat([a["_deformation_table"], new_table], -1)
    a["xyz_gradient_accum"] = torch.zeros(n, 1)
    a["_deformation_accum"] = torch.zeros(n, 3)
    a["denom"] = torch.zeros(n, 1)
    a["max_radii2D"] = torch.zeros(n)


def prune_points(state, mask):
    """prune_points + _prune_optimizer (scene/gaussian_model.py:1044-1089): rows with mask=True are removed from
    the parameters, their Adam moments and every per-splat statistic."""
    keep = ~mask
    for g in GROUPS:
        state["params"][g] = state["params"][g][keep]
        if g in state["exp_avg"]:
            state["exp_avg"][g] = state["exp_avg"][g][keep]
            state["exp_avg_sq"][g] = state["exp_avg_sq"][g][keep]
    for a in AUX:
        state["aux"][a] = state["aux"][a][keep]


def densify_and_clone(state, grads, grad_threshold, scene_extent, percent_dense=0.01):
    """scene/gaussian_model.py:1480-1506: small splats with a large mean view-space gradient are duplicated."""
    p = state["params"]
    sel = torch.norm(grads, dim=-1) >= grad_threshold
    sel = torch.logical_and(sel, torch.max(torch.exp(p["scaling"]), dim=1).values <= percent_dense * scene_extent)
    _append(state, {g: p[g][sel] for g in GROUPS}, state["aux"]["_deformation_table"][sel])
    return sel


def build_rotation(r):
    """utils/general_utils.py:85-106."""
    q = r / torch.sqrt((r * r).sum(1))[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.zeros(q.shape[0], 3, 3)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - w * z)
    R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y)
    R[:, 2, 1] = 2 * (y * z + w * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def densify_and_splitv2(state, grads, grad_threshold, scene_extent, N=2, percent_dense=0.01, samples=None):
    """scene/gaussian_model.py:1207-1244: large splats with a large mean gradient are replaced by N samples drawn
    from themselves (scale / (0.8 N)).  `grads` may be shorter than the table (rows appended by the clone step
    count as zero).  `samples` [N*n_sel, 3]: the N(0, scale) draws, when the caller wants to fix them."""
    p = state["params"]
    n = p["xyz"].shape[0]
    padded = torch.zeros(n)
    padded[:grads.shape[0]] = grads.squeeze()
    scale = torch.exp(p["scaling"])
    sel = torch.logical_and(padded >= grad_threshold, torch.max(scale, dim=1).values > percent_dense * scene_extent)
    stds = scale[sel].repeat(N, 1)
    if samples is None:
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds)
    rots = build_rotation(p["rotation"][sel]).repeat(N, 1, 1)
    new = {g: p[g][sel].repeat(N, *([1] * (p[g].dim() - 1))) for g in GROUPS}
    new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + p["xyz"][sel].repeat(N, 1)
    new["scaling"] = torch.log(scale[sel].repeat(N, 1) / (0.8 * N))
    n_sel = int(sel.sum())
    _append(state, new, state["aux"]["_deformation_table"][sel].repeat(N))
    prune_points(state, torch.cat((sel, torch.zeros(N * n_sel, dtype=torch.bool))))
    return sel


def reset_opacity(state):
    """scene/gaussian_model.py:897-903 + replace_tensor_to_optimizer :1029-1042."""
    op = torch.sigmoid(state["params"]["opacity"])
    x = torch.min(op, torch.ones_like(op) * 0.01)
    state["params"]["opacity"] = torch.log(x / (1 - x))
    state["exp_avg"]["opacity"] = torch.zeros_like(op)
    state["exp_avg_sq"]["opacity"] = torch.zeros_like(op)


def state_from_fixture(fx, tag):
    st = {"params": {}, "exp_avg": {}, "exp_avg_sq": {}, "aux": {}}
    for g in GROUPS:
        st["params"][g] = torch.from_numpy(fx[f"{tag}.{g}"]).clone()
        if f"{tag}.{g}.exp_avg" in fx:
            st["exp_avg"][g] = torch.from_numpy(fx[f"{tag}.{g}.exp_avg"]).clone()
            st["exp_avg_sq"][g] = torch.from_numpy(fx[f"{tag}.{g}.exp_avg_sq"]).clone()
    for a in AUX:
        st["aux"][a] = torch.from_numpy(fx[f"{tag}.{a}"]).clone()
    return st
