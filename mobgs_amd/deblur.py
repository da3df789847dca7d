"""The blurry-view part of one training iteration (reference: /root/reference/train.py:430-541), sharded.

For every view of the batch the reference renders the view's own camera in train mode (`get_static=True,
get_dynamic=True`, :441-443), asks `blcekernel.get_warped_cams()` for K = 9 latent cameras and exposure offsets
(:472), renders the 8 latent frames that are not the mid one (`latent_sharp_id != half`, :510-518; for the mid slot
it re-uses the train-mode render, :507-509) and averages: `pred = mean(rendered_images) + 1e-10` (:540-541).

`render_blurry_batch` does the same for the (view, sub-frame) units THIS rank owns (mobgs_amd.distributed) and
exchanges the partial sums once for the whole batch.  With world = 1 it is exactly the reference's loop.
bench.py times it (the "K-sub-frame deblur throughput" of BASELINE.json), tests/test_gpu_config4.py checks it against
the oracle, examples/train_synth.py trains with it.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .distributed import SubframeShard
from .gaussian_renderer import render


def render_blurry_batch(cams: Sequence, stat_pc, dyn_pc, bg_color: torch.Tensor, shard: SubframeShard,
                        blce=None, n_sub: int = 9, exposures: Optional[Sequence[Sequence]] = None,
                        train_mode_mid: bool = True, pipe=None) -> Tuple[torch.Tensor, Dict[int, dict]]:
    """cams: the batch's view cameras.  blce: a mobgs_amd.blce.blceKernel (None: every sub-frame uses the view's own
    camera and `exposures[v][k]` / 0 as exposure offset -- the reference before `start_warp`).
    -> (pred [V,3,H,W] on every rank, {view index: result dict of its mid (train-mode) render} for the mid frames this
    rank rendered)."""
    V = len(cams)
    half = n_sub // 2
    mine = shard.view_units(V, n_sub)
    warped: Dict[int, Tuple[List, torch.Tensor]] = {}
    mids: Dict[int, dict] = {}
    like = None

    def latent(v):
        if v not in warped:
            if blce is not None:
                warped[v] = blce.get_warped_cams(cams[v], None, None)  # train.py:472 (fwd/bwd cams are unused there)
            else:
                e = exposures[v] if exposures is not None else [0] * n_sub
                warped[v] = ([cams[v]] * n_sub, e)
        return warped[v]

    def unit(v, k):
        nonlocal like
        if k == half:
            pkg = render(cams[v], stat_pc, dyn_pc, pipe, bg_color, get_static=train_mode_mid,
                         get_dynamic=train_mode_mid)
            mids[v] = pkg
        else:
            wc, expo = latent(v)
            d = expo[k]
            pkg = render(wc[k], stat_pc, dyn_pc, pipe, bg_color, get_static=True, get_dynamic=True, delta_exposure=d)
        like = pkg["render"]
        return pkg["render"]

    if not mine:  # more ranks than units: contribute zeros (image size from the first camera)
        c = cams[0]
        like = torch.zeros(3, int(c.image_height), int(c.image_width), device=bg_color.device)
        return shard.render_blurry_views(unit, V, n_sub, like=like), mids
    # the image shape is known after the first unit; render_blurry_views only needs `like` for views this rank has no
    # unit of, so hand it a lazily-filled zero image
    first_v, first_k = mine[0]
    img0 = unit(first_v, first_k)
    cache = {(first_v, first_k): img0}

    def unit_cached(v, k):
        return cache.pop((v, k)) if (v, k) in cache else unit(v, k)

    return shard.render_blurry_views(unit_cached, V, n_sub, like=torch.zeros_like(img0)), mids
