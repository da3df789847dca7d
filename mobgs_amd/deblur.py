"""The blurry-view part of one training iteration (reference: /root/reference/train.py:430-541), sharded.

For every view of the batch the reference renders the view's own camera in train mode (`get_static=True,
get_dynamic=True`, :441-443), asks `blcekernel.get_warped_cams()` for K = 9 latent cameras and exposure offsets
(:472), renders the 8 latent frames that are not the mid one (`latent_sharp_id != half`, :510-518; for the mid slot
it re-uses the train-mode render, :507-509) and averages: `pred = mean(rendered_images) + 1e-10` (:540-541).

`render_blurry_batch` does the same for the (view, sub-frame) units THIS rank owns (mobgs_amd.distributed) and
exchanges the partial sums once for the whole batch.  With world = 1 it is exactly the reference's loop.
bench.py times it (the "K-sub-frame deblur throughput" of BASELINE.json), tests/test_gpu_config4.py checks it against
the oracle, examples/train_deblur_synth.py trains with it.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .distributed import SubframeShard
from .gaussian_renderer import render


def get_flow_batch(cams: Sequence, stat_pc, dyn_pc, bg_color: torch.Tensor, shard: SubframeShard, n_sub: int = 9,
                   exposure_max_delta: float = 1.0, pipe=None, weighted: bool = False) -> Dict[Tuple[int, int], tuple]:
    """The n_sub get_flow() calls per view of train.py:564-579 (exposure offsets exposure_max_delta * (k - half) /
    half), for the (view, sub-frame) units THIS rank owns: {(view, k): (exp2mid, mid2exp, latent_img, latent_alpha)}.
    The units are dealt round-robin like the renders, rotated by the number of render units so that the ranks the
    renders left with one unit fewer take the surplus here; a rank's calls of one view share the mid-exposure state
    (get_flow_many).  The flow loss is a sum of per-sub-frame terms: the owner of a unit forms the unit's term, the
    gradient SUM of all_reduce_gradients() then counts every term once -- and because those terms also read the
    blurry prediction, the prediction's exchange must reduce its backward (render_blurry_batch(...,
    rank_local_terms=True)) and terms all ranks form identically on it go through shard.replicated_term()."""
    from .gaussian_renderer import get_flow_many
    V, half = len(cams), n_sub // 2
    mine = shard.planned_units(shard.iteration_plan(V, n_sub)["flow"], n_sub) if weighted \
        else shard.view_units(V, n_sub, offset=V * n_sub)
    out: Dict[Tuple[int, int], tuple] = {}
    for v in sorted({v for v, _ in mine}):
        ks = [k for vv, k in mine if vv == v]
        deltas = [exposure_max_delta * (k - half) / half for k in ks]
        for k, res in zip(ks, get_flow_many(cams[v], stat_pc, dyn_pc, pipe, bg_color, deltas)):
            out[(v, k)] = res
    return out


def render_blurry_batch(cams: Sequence, stat_pc, dyn_pc, bg_color: torch.Tensor, shard: SubframeShard,
                        blce=None, n_sub: int = 9, exposures: Optional[Sequence[Sequence]] = None,
                        train_mode_mid: bool = True, pipe=None, rank_local_terms: bool = False,
                        weighted: bool = False, with_flows: bool = False, overlap: bool = False,
                        batched_latent: bool = True, as_list: bool = False) -> Tuple[torch.Tensor, Dict[int, dict]]:
    """cams: the batch's view cameras.  blce: a mobgs_amd.blce.blceKernel (None: every sub-frame uses the view's own
    camera and `exposures[v][k]` / 0 as exposure offset -- the reference before `start_warp`).
    -> (pred [V,3,H,W] on every rank, {view index: result dict of its mid (train-mode) render} for the mid frames this
    rank rendered).  rank_local_terms: some loss term on `pred` is formed by one rank only (get_flow_batch): the
    exchange then all-reduces its backward too.  as_list: pred as a list of V [3,H,W] tensors that share no autograd node
    (SubframeShard.backward_by_view back-propagates one view at a time)."""
    V = len(cams)
    half = n_sub // 2
    # weighted: units dealt by cost (SubframeShard.iteration_plan: the train-mode mid frames weigh 2.3 latent renders;
    # with_flows: the get_flow units of the same iteration -- get_flow_batch(weighted=True) -- share the pool);
    # overlap: one asynchronous image all-reduce per view instead of one for the batch
    mine = shard.planned_units(shard.iteration_plan(V, n_sub, with_flows)["render"], n_sub) if weighted \
        else shard.view_units(V, n_sub)
    warped: Dict[int, Tuple[List, torch.Tensor]] = {}
    mids: Dict[int, dict] = {}
    like = None

    def latent(v):
        if v not in warped:
            if blce is not None:
                wc, expo = blce.get_warped_cams(cams[v], None, None)  # train.py:472 (fwd/bwd cams are unused there)
                if exposures is not None:
                    # the caller's offsets override BLCE's: train.py:503-506 renders the warped cameras with
                    # delta_exposure = 0 while start_warp < iteration <= start_warp_dynamic
                    expo = exposures[v]
                warped[v] = (wc, expo)
            else:
                e = exposures[v] if exposures is not None else [0] * n_sub
                warped[v] = ([cams[v]] * n_sub, e)
        return warped[v]

    # batched_latent: this rank's latent (non-mid) sub-frames of a view go through ONE render_many() call -- one
    # projection / binning / sort / compositing pass over K' cameras with per-camera geometry instead of K' render()
    # calls (gaussian_renderer.render_many: the win is at small image sizes, where a single render is latency-bound)
    batch_cache: Dict[Tuple[int, int], torch.Tensor] = {}

    def latent_batch(v):
        from .gaussian_renderer import render_many
        ks = [k for vv, k in mine if vv == v and k != half]
        wc, expo = latent(v)
        outs = render_many([wc[k] for k in ks], stat_pc, dyn_pc, pipe, bg_color, [expo[k] for k in ks])
        for k, o in zip(ks, outs):
            batch_cache[(v, k)] = o["render"]

    def unit(v, k):
        nonlocal like
        if k == half:
            pkg = render(cams[v], stat_pc, dyn_pc, pipe, bg_color, get_static=train_mode_mid,
                         get_dynamic=train_mode_mid)
            mids[v] = pkg
            img = pkg["render"]
        elif batched_latent:
            if (v, k) not in batch_cache:
                latent_batch(v)
            img = batch_cache.pop((v, k))
        else:
            wc, expo = latent(v)
            d = expo[k]
            img = render(wc[k], stat_pc, dyn_pc, pipe, bg_color, get_static=True, get_dynamic=True,
                         delta_exposure=d)["render"]
        like = img
        return img

    if not mine:  # more ranks than units: contribute zeros (image size from the first camera)
        c = cams[0]
        like = torch.zeros(3, int(c.image_height), int(c.image_width), device=bg_color.device)
        return shard.render_blurry_views(unit, V, n_sub, like=like, reduce_backward=rank_local_terms, units=mine,
                                         overlap=overlap, as_list=as_list), mids
    # the image shape is known after the first unit; render_blurry_views only needs `like` for views this rank has no
    # unit of, so hand it a lazily-filled zero image
    first_v, first_k = mine[0]
    img0 = unit(first_v, first_k)
    cache = {(first_v, first_k): img0}

    def unit_cached(v, k):
        return cache.pop((v, k)) if (v, k) in cache else unit(v, k)

    return shard.render_blurry_views(unit_cached, V, n_sub, like=torch.zeros_like(img0),
                                     reduce_backward=rank_local_terms, units=mine, overlap=overlap, as_list=as_list), mids
