#!/usr/bin/env python
"""A miniature of the reference's DEBLUR training iteration (train.py:430-700) on a synthetic scene, end to end on the
MI355X path and, under torchrun, sharded over GPUs:

  per view of the batch   blcekernel.get_warped_cams -> K = 9 latent renders (mid frame in train mode) -> mean = blurry
                          prediction (mobgs_amd.deblur.render_blurry_batch: ONE all-reduce for the batch)
                          K get_flow() calls (mobgs_amd.deblur.get_flow_batch, sharded like the renders)
  loss                    photometric (L1 + 0.2 D-SSIM, fused) on the prediction, depth / mask terms on the mid render,
                          the flow-consistency term of train.py:651-671 on the get_flow outputs (both grid_sample
                          warps + both masked L1 terms: mobgs_amd.loss_utils.flow_warp_loss, one fused kernel each
                          way; sharded runs: a rank-local stand-in per flow unit), a regulariser on the scales
  backward                ops.LeafGradSink + distributed.FlatGradients, ONE in-place gradient all-reduce that also carries
                          the mid-frame densification statistics
  step                    Adam on both Gaussian sets, the decoder and the BLCE parameters; densification statistics

    python examples/train_deblur_synth.py [--iters 60]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/train_deblur_synth.py

The "ground truth" blurry views are K-frame averages of the same scene with perturbed colours seen through jittered
cameras, so the loss has somewhere to go.  Used by tests/test_gpu_train_loop.py as an integration test.
"""
import argparse
import os
import sys

import torch

torch.autograd.set_multithreading_enabled(False)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobgs_amd.blce import blceKernel  # noqa: E402
from mobgs_amd.camera import PinholeCamera  # noqa: E402
from mobgs_amd.deblur import get_flow_batch, render_blurry_batch  # noqa: E402
from mobgs_amd.densify import TrainableGaussians  # noqa: E402
from mobgs_amd.distributed import FlatGradients, SubframeShard  # noqa: E402
from mobgs_amd.gaussian_renderer import render  # noqa: E402
from mobgs_amd.helper_model import Sandwich  # noqa: E402
from mobgs_amd.loss_utils import flow_warp_loss, l1_loss, photometric_loss  # noqa: E402
from mobgs_amd.ops import LeafGradSink  # noqa: E402
from mobgs_amd.optim import fused_adam_step  # noqa: E402
from mobgs_amd.synth import SynthCamera, dynamic_extras, gaussian_cloud  # noqa: E402
from train_synth import Opt  # noqa: E402  (examples/ is on sys.path when run as a script or from the test)

K = 9


def view_pose(i):
    w2c = torch.eye(4)
    w2c[:3, 3] = torch.tensor([0.04 * i, -0.02 * i, 0.03 * i])
    return w2c


class _Intrinsics:
    """The five numbers main_utils.get_normals reads from a dycheck camera (main_utils.py:95-141)."""

    def __init__(self, K):
        self.scale_factor_x, self.scale_factor_y = float(K[0, 0]), float(K[1, 1])
        self.principal_point_x, self.principal_point_y = float(K[0, 2]), float(K[1, 2])
        self.skew = float(K[0, 1])


class DeblurTrainer:
    """State of the miniature training loop; `iteration()` is ONE iteration of train.py:430-807 on this stack:
    blurry views (K latent renders each, BLCE cameras) -> K get_flow() calls per view -> photometric + depth + mask +
    normal + flow + scale terms -> backward into the flat gradient buffer -> (all-reduce) -> densification statistics ->
    Adam on both Gaussian sets (incl. the decoder) and the BLCE parameters.  bench.py times it at full size."""

    def __init__(self, dev="cuda:0", ns=4000, nd=2000, width=256, height=192, n_views=2, seed=0, lambda_flow=1e-2,
                 shard=None, iters=40):
        from mobgs_amd.main_utils import get_normals
        self.get_normals = get_normals
        self.shard = shard = shard or SubframeShard()
        self.ns, self.nd, self.n_views, self.lambda_flow = ns, nd, n_views, lambda_flow
        scam = SynthCamera().scaled(width, height)
        torch.manual_seed(seed)
        dec = Sandwich(9, 3).to(dev)
        sp, dp = gaussian_cloud(ns, scam, seed), gaussian_cloud(nd, scam, seed + 1)
        dx = dynamic_extras(dp["xyz"], seed)
        self.stat = stat = TrainableGaussians(sp, None, dec, device=dev)
        self.dyn = dyn = TrainableGaussians(dp, dx, dec, device=dev)
        self.bg = bg = torch.zeros(9, device=dev)
        g = torch.Generator().manual_seed(seed + 100)
        self.cams = cams = []
        for i in range(n_views):
            c = PinholeCamera(width, height, scam.K, view_pose(i), time=(5.0 + 6 * i) / 23.0, max_time=23, device=dev)
            c.uid = i
            cams.append(c)
        self.meta = _Intrinsics(scam.K)
        # blurry targets: mean of K renders of a colour-shifted copy of the scene at spread-out exposure offsets
        with torch.no_grad():
            tsp = dict(sp, features_dc=sp["features_dc"] + 0.4 * torch.randn(sp["features_dc"].shape, generator=g))
            tdp = dict(dp, features_dc=dp["features_dc"] + 0.4 * torch.randn(dp["features_dc"].shape, generator=g))
            tstat, tdyn = TrainableGaussians(tsp, None, dec, device=dev), TrainableGaussians(tdp, dx, dec, device=dev)
            self.targets, self.depths, self.normals = [], [], []
            for c in cams:
                outs = [render(c, tstat, tdyn, None, bg, delta_exposure=float(d))
                        for d in torch.linspace(-0.6, 0.6, K)]
                self.targets.append(torch.stack([o["render"] for o in outs]).mean(0).clamp(0, 1))
                self.depths.append(outs[K // 2]["depth"].detach())
                self.normals.append(get_normals(self.depths[-1] + 1e-6, self.meta).detach())
                c.image = self.targets[-1]  # BLCE's blur statistic reads the (blurry) input image of the view
            del tstat, tdyn
        self.gt = torch.stack(self.targets)
        torch.manual_seed(seed + 1)
        self.blce = blceKernel(num_views=n_views, num_warp=K, iteration=max(iters, 1)).to(dev)
        self.opt = opt = Opt()
        stat.training_setup(opt)
        dyn.training_setup(opt)
        dyn.optimizer.param_groups = [gr for gr in dyn.optimizer.param_groups if gr["name"] != "decoder"]
        params = [p for gr in stat.optimizer.param_groups + dyn.optimizer.param_groups for p in gr["params"]] \
            + list(self.blce.model.get_params())
        self.bucket = FlatGradients(params, extra={f"view{v}": 3 * (ns + nd) for v in range(n_views)})
        # N > 1: one gradient message per view, exchanged while the next view back-propagates (backward_by_view)
        self.view_buckets = ([FlatGradients(params, extra={f"view{v}": 3 * (ns + nd)}) for v in range(n_views)]
                             if shard.collective else None)

    def _iteration_sharded(self) -> torch.Tensor:
        """N > 1: the same iteration with its loss kept apart per view, so that the backward pass runs view by view and
        each view's gradient message (with its densification statistics) is all-reduced on the communication stream while
        the next view back-propagates (SubframeShard.backward_by_view)."""
        shard, stat, dyn, blce, ns, V = self.shard, self.stat, self.dyn, self.blce, self.ns, self.n_views
        preds, mids = render_blurry_batch(self.cams, stat, dyn, self.bg, shard, blce=blce, n_sub=K, rank_local_terms=True,
                                          weighted=True, with_flows=True, overlap=True, as_list=True)
        flows = get_flow_batch(self.cams, stat, dyn, self.bg, shard, n_sub=K, weighted=True)
        photos = [photometric_loss(preds[v][None], self.gt[v][None], self.opt.lambda_dssim) for v in range(V)]

        def view_backward(v):
            loss = shard.replicated_term(photos[v]) / V          # the mean over V equal-sized images, view by view
            if v in mids:
                pkg = mids[v]
                for key in ("s_render", "s_depth", "d_alpha", "d_depth", "s_alpha"):
                    pkg[key]
                normal = self.get_normals(pkg["depth"] + 1e-6, self.meta)
                loss = loss + 0.05 * l1_loss(pkg["depth"], self.depths[v]) + 0.01 * pkg["d_alpha"].mean() \
                    + 0.01 * l1_loss(normal, self.normals[v])
            if self.lambda_flow != 0:
                for (vv, k), (e2m, m2e, limg, lalpha) in flows.items():
                    if vv == v:
                        loss = loss + self.lambda_flow / K * (l1_loss(limg, preds[v]) + 1e-3 * (e2m - m2e).abs().mean()
                                                              + 0.1 * lalpha.mean())
            if v == 0:
                loss = loss + shard.replicated_term(1e-4 * ((stat._scaling ** 2).mean() + (dyn._scaling ** 2).mean()))
            if loss.requires_grad:
                with LeafGradSink(stat, dyn, extra=blce.model.get_params()):
                    loss.backward()

        def after_view(v):
            if v in mids:
                shard.put_densification_stats(self.view_buckets[v], f"view{v}", mids[v]["viewspace_points"].grad,
                                              mids[v]["radii"])

        shard.backward_by_view(self.view_buckets, view_backward, after_view)
        with torch.no_grad():
            for v in range(V):
                grad2d, radii = shard.get_densification_stats(self.view_buckets[v], f"view{v}")
                vis = radii > 0
                stat.add_densification_stats(grad2d[:ns], vis[:ns], radii=radii[:ns])
                dyn.add_densification_stats(grad2d[ns:], vis[ns:], radii=radii[ns:])
        fused_adam_step([stat.optimizer, dyn.optimizer, blce.optimizer])
        return torch.stack([p.detach() for p in photos]).mean()

    def iteration(self) -> torch.Tensor:
        if self.shard.collective:   # N > 1 (or a forced one-rank group: the same code path on one GPU)
            return self._iteration_sharded()
        photo = self.forward_backward()
        self.optimizer_step()
        return photo

    def optimizer_step(self):
        # train.py:790-807 steps the three optimisers one after the other (~26 one-tensor groups x 8 launches); here
        # ONE launch performs the same Adam update on all of them (mobgs_amd.optim)
        fused_adam_step([self.stat.optimizer, self.dyn.optimizer, self.blce.optimizer])

    def forward_backward(self) -> torch.Tensor:
        """Everything of the single-process iteration but the optimiser step: renders, flows, loss, backward into the flat
        gradient buffer, densification statistics -- device work over persistent tensors only, i.e. what
        mobgs_amd.graphed.GraphedCallable can record once and replay (scripts/bench_small_scene_iteration.py --graph)."""
        shard, stat, dyn, blce, bucket, ns = self.shard, self.stat, self.dyn, self.blce, self.bucket, self.ns
        bucket.zero()
        multi = shard.world > 1  # units of both families dealt by cost, per-view asynchronous image exchange
        pred, mids = render_blurry_batch(self.cams, stat, dyn, self.bg, shard, blce=blce, n_sub=K,
                                         rank_local_terms=True, weighted=multi, with_flows=True, overlap=multi)
        flows = get_flow_batch(self.cams, stat, dyn, self.bg, shard, n_sub=K, weighted=multi)
        photo = photometric_loss(pred, self.gt, self.opt.lambda_dssim)
        loss = shard.replicated_term(photo)                      # every rank forms it on the replicated prediction
        for v, pkg in mids.items():                              # the rank that rendered the mid frame
            for key in ("s_render", "s_depth", "d_alpha", "d_depth", "s_alpha"):
                pkg[key]   # train.py:445-464 reads these five auxiliary images of the mid render every iteration
            normal = self.get_normals(pkg["depth"] + 1e-6, self.meta)   # train.py:590
            loss = loss + 0.05 * l1_loss(pkg["depth"], self.depths[v]) + 0.01 * pkg["d_alpha"].mean() \
                + 0.01 * l1_loss(normal, self.normals[v])
        if not multi and self.lambda_flow != 0:
            # train.py:608-617 + :651-671: the flow-consistency term on the tensors the reference concatenates --
            # both grid_sample warps and both masked L1 terms in one forward and one backward kernel
            views = range(self.n_views)
            cat = lambda i: torch.stack([torch.cat([flows[(v, k)][i] for k in range(K)]) for v in views])  # noqa: E731
            loss = loss + flow_warp_loss(torch.stack([mids[v]["render"] for v in views]),
                                         torch.stack([torch.stack([flows[(v, k)][2] for k in range(K)]) for v in views]),
                                         cat(0), cat(1), cat(3).unsqueeze(2),
                                         torch.stack([mids[v]["d_alpha"].reshape(1, *pred.shape[-2:]) for v in views]),
                                         self.lambda_flow)
        elif self.lambda_flow != 0:   # (weight 0, the shipped seesaw / children configs: the calls above are made, as
            # train.py makes them, and their results stay out of the loss graph -- flow_warp_loss returns a constant)
            # flow units sharded over ranks: each owner forms a rank-local stand-in (the reference's term normalises
            # over all (view, exposure) pairs and samples the view's mid render, which lives on one rank)
            for (v, k), (e2m, m2e, limg, lalpha) in flows.items():
                loss = loss + self.lambda_flow / K * (l1_loss(limg, pred[v]) + 1e-3 * (e2m - m2e).abs().mean()
                                                      + 0.1 * lalpha.mean())
        loss = loss + shard.replicated_term(1e-4 * ((stat._scaling ** 2).mean() + (dyn._scaling ** 2).mean()))
        with LeafGradSink(stat, dyn, extra=blce.model.get_params()):
            loss.backward()
        for v, pkg in mids.items():
            shard.put_densification_stats(bucket, f"view{v}", pkg["viewspace_points"].grad, pkg["radii"])
        shard.all_reduce_gradients(bucket)
        with torch.no_grad():
            for v in range(self.n_views):
                grad2d, radii = shard.get_densification_stats(bucket, f"view{v}")
                vis = radii > 0
                stat.add_densification_stats(grad2d[:ns], vis[:ns], radii=radii[:ns])
                dyn.add_densification_stats(grad2d[ns:], vis[ns:], radii=radii[ns:])
        return photo.detach()


    def iteration_unchanged(self) -> torch.Tensor:
        """The SAME iteration written the way /root/reference/train.py:430-807 writes it -- what runs after nothing but
        the import swap of INTEGRATION.md section 1: one render() per latent sub-frame, every one of them with
        get_static = get_dynamic = True (train.py:441 and :512), one get_flow() per exposure (:570-579), l1_loss + ssim
        as two calls (:621-628), the flow-consistency term as torch statements (two F.grid_sample + two masked l1_loss,
        :651-671), loss.backward() into ordinary .grad tensors, viewspace_points.grad for the densification statistics
        (:634-648) and the three optimizer.step() calls (:790-807; the optimisers are optim.FusedAdam, the torch.optim.Adam
        subclass TrainableGaussians.training_setup / blceKernel build: one launch per step() instead of ~8 per parameter group).  None of the opt-in entry points (render_many,
        get_flow_many, flow_warp_loss, fused_adam_step, LeafGradSink, FlatGradients).  Single process."""
        import torch.nn.functional as F
        from mobgs_amd.gaussian_renderer import get_flow
        from mobgs_amd.loss_utils import ssim
        stat, dyn, blce, ns, bg = self.stat, self.dyn, self.blce, self.ns, self.bg
        opts = [stat.optimizer, dyn.optimizer, blce.optimizer]
        for o in opts:
            o.zero_grad(set_to_none=True)
        images, ori, d_alphas, depths, normals, vsp, radii = [], [], [], [], [], [], []
        lat_img, lat_alpha, e2m_all, m2e_all = [], [], [], []
        for cam in self.cams:
            pkg = render(cam, stat, dyn, None, bg, get_static=True, get_dynamic=True)                      # train.py:441
            for key in ("s_render", "s_depth", "d_alpha", "d_depth", "s_alpha"):
                pkg[key]                                                                                    # :445-464
            image_ori = pkg["render"]
            warped_cams, exposure_time = blce.get_warped_cams(cam, None, None)                             # :472
            half = len(warped_cams) // 2
            rendered = []
            for k, wc in enumerate(warped_cams):                                                           # :502-518
                if k == half:
                    rendered.append(image_ori)
                else:
                    rendered.append(render(wc, stat, dyn, None, bg, get_static=True, get_dynamic=True,
                                           delta_exposure=exposure_time[k])["render"])
            images.append(torch.mean(torch.stack(rendered, dim=0), dim=0) + 1e-10)                         # :540-541
            cur = [[], [], [], []]
            for k in range(len(warped_cams)):                                                              # :570-579
                outs = get_flow(cam, stat, dyn, None, bg, delta_exposure=1.0 * (k - half) / half)
                for lst, o in zip(cur, outs):
                    lst.append(o)
            e2m_all.append(torch.cat(cur[0], 0).unsqueeze(0))
            m2e_all.append(torch.cat(cur[1], 0).unsqueeze(0))
            lat_img.append(torch.cat([t.unsqueeze(0) for t in cur[2]], 0).unsqueeze(0))
            lat_alpha.append(torch.cat([t.unsqueeze(0) for t in cur[3]], 0).unsqueeze(0))
            ori.append(image_ori.unsqueeze(0))
            d_alphas.append(pkg["d_alpha"].unsqueeze(0))
            depths.append(pkg["depth"])
            normals.append(self.get_normals(pkg["depth"] + 1e-6, self.meta))                               # :590
            vsp.append(pkg["viewspace_points"])
            radii.append(pkg["radii"])
        pred = torch.stack(images)
        Ll1 = l1_loss(pred, self.gt)                                                                       # :621-628
        photo = (1.0 - self.opt.lambda_dssim) * Ll1 + self.opt.lambda_dssim * (1.0 - ssim(pred, self.gt))
        loss = photo
        for v in range(self.n_views):
            loss = loss + 0.05 * l1_loss(depths[v], self.depths[v]) + 0.01 * d_alphas[v].mean() \
                + 0.01 * l1_loss(normals[v], self.normals[v])
        if self.lambda_flow != 0:                                                                          # :651-671
            H, W = pred.shape[-2:]
            E = K
            ori_t, lat_t = torch.cat(ori, 0), torch.cat(lat_img, 0)
            la_t, da_t = torch.cat(lat_alpha, 0), torch.cat(d_alphas, 0).reshape(self.n_views, 1, H, W)

            def norm(c):
                c = torch.stack([c[..., 0] / (W - 1), c[..., 1] / (H - 1)], -1)
                return (2.0 * c - 1.0).flatten(0, 1)
            ori_e = ori_t.unsqueeze(1).expand(-1, E, -1, -1, -1).flatten(0, 1)
            w_e2m = F.grid_sample(ori_e, norm(torch.cat(e2m_all, 0)), mode="bilinear", padding_mode="border",
                                  align_corners=False).reshape(-1, E, 3, H, W)
            w_m2e = F.grid_sample(lat_t.flatten(0, 1), norm(torch.cat(m2e_all, 0)), mode="bilinear", padding_mode="border",
                                  align_corners=False).reshape(-1, E, 3, H, W)
            loss = loss + self.lambda_flow * (
                l1_loss(w_e2m.flatten(0, 1), lat_t.flatten(0, 1), mask=la_t.flatten(0, 1))
                + l1_loss(w_m2e.flatten(0, 1), ori_e,
                          mask=da_t.unsqueeze(1).expand(-1, E, -1, -1, -1).flatten(0, 1)))
        loss = loss + 1e-4 * ((stat._scaling ** 2).mean() + (dyn._scaling ** 2).mean())
        loss.backward()
        with torch.no_grad():                                                                              # :634-648
            for v in range(self.n_views):
                grad2d = vsp[v].grad.squeeze(0)
                vis = radii[v] > 0
                stat.add_densification_stats(grad2d[:ns], vis[:ns], radii=radii[v][:ns])
                dyn.add_densification_stats(grad2d[ns:], vis[ns:], radii=radii[v][ns:])
        for o in opts:                                                                                     # :790-807
            o.step()
        return photo.detach()


def train(dev="cuda:0", iters=40, ns=4000, nd=2000, width=256, height=192, n_views=2, seed=0, lambda_flow=1e-2,
          shard=None, log=None, graph=False):
    """graph=True (single process): forward + losses + backward of the iteration recorded ONCE as a HIP graph
    (mobgs_amd.graphed.GraphedCallable) and replayed, the Adam step outside -- for small images / few Gaussians, where an
    iteration is ~1400 launches and bound by the host (the reference's own 512x288 / 30 k operating point: 9.5 -> 7.5 ms)."""
    t = DeblurTrainer(dev, ns, nd, width, height, n_views, seed, lambda_flow, shard, iters)
    history = []
    fb, pending = None, []
    for it in range(1, iters + 1):
        if graph and it == 2 and not t.shard.collective:   # (iteration 1 ran eagerly: arenas and hints exist)
            from mobgs_amd.graphed import GraphedCallable
            fb = GraphedCallable(t.forward_backward, warmup=0)
        if fb is not None:
            photo = fb()
            t.optimizer_step()
            # the counts of the replays land in pinned rows: check() sees the replays that have COMPLETED.  The host enqueues a
            # replay in a fraction of its run time, so it is kept at most two iterations ahead (an event per iteration) -- an
            # arena outgrown by the moving scene (its frame saw empty lists: a background image, no splat gradients) is then
            # noticed two iterations late at most, and the iteration recorded again
            ev = torch.cuda.Event()
            ev.record()
            pending.append(ev)
            if len(pending) > 2:
                pending.pop(0).synchronize()
            if not fb.check():
                torch.cuda.synchronize()
                fb.recapture()
            history.append(float(photo))
        else:
            history.append(float(t.iteration()))
        if it == 2:   # everything long-lived exists now: keep Python's cyclic collector off it (66 ms per generation-2
            import gc  # pass over a torch process's objects on this host, in the middle of an iteration: DESIGN section 5)
            gc.collect()
            gc.freeze()
        if log and (it % log == 0 or it == 1) and t.shard.rank == 0:
            print(f"it {it:4d}  photometric {history[-1]:.5f}")
    return history, t.stat, t.dyn, t.blce, t.bucket


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--ns", type=int, default=20000)
    ap.add_argument("--nd", type=int, default=10000)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=288)
    ap.add_argument("--graph", action="store_true", help="replay forward + backward as one HIP graph (single process)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    train(dev=f"cuda:{local}", iters=a.iters, ns=a.ns, nd=a.nd, width=a.width, height=a.height, log=10, graph=a.graph)
