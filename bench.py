#!/usr/bin/env python
"""Benchmark of the MoBGS render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): train-step renders/sec -- one "step" on one GPU is one full `render()` forward+backward
of BASELINE config #2: the synthetic "seesaw" scene of SURVEY.md section 8d (200k static + 100k dynamic Gaussians,
1352x1014, lean mode as eval.py:125), i.e. per-splat prep (Hermite spline, activations) -> projection -> tile
lists + per-tile depth sort -> 10-channel compositing -> expected depth + colour decoder, and the backward pass
to every Gaussian leaf, the decoder weights and the camera matrix.  Inputs are resident in HBM before the timed
region.  With N > 1 every rank renders a different latent sub-frame of one blurry view (weak scaling: per-GPU
work fixed), the partial images are summed with an RCCL all-reduce into the blurry prediction, and the
parameter gradients are all-reduced as one flat buffer (train.py:502-541 sharded as SURVEY.md section 8e).

Prints ONE JSON line on rank 0 (see the task contract), including
  roofline     -- the dominant kernel (raster_bwd): algorithmic bytes / HIP-event time vs the 8 TB/s HBM peak
  cpu_baseline -- oracle/gsplat_cpu.c (OpenMP port of upstream's kernels) timed on this host, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mobgs_amd import profiler  # noqa: E402
from mobgs_amd.camera import PinholeCamera  # noqa: E402
from mobgs_amd.gaussian_model import GaussianParams  # noqa: E402
from mobgs_amd.helper_model import Sandwich  # noqa: E402
from mobgs_amd.synth import SynthCamera, dynamic_extras, gaussian_cloud  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_scene(dev, ns, nd, width, height, seed=0):
    scam = SynthCamera().scaled(width, height) if (width, height) != (1352, 1014) else SynthCamera()
    stat_p = gaussian_cloud(ns, scam, seed)
    dyn_p = gaussian_cloud(nd, scam, seed + 1)
    dyn_x = dynamic_extras(dyn_p["xyz"], seed)
    torch.manual_seed(seed)
    dec = Sandwich(9, 3).to(dev)
    stat = GaussianParams(stat_p, None, dec, dev, requires_grad=True)
    dyn = GaussianParams(dyn_p, dyn_x, dec, dev, requires_grad=True)
    cam = PinholeCamera(width, height, scam.K, torch.eye(4), time=scam.time, max_time=scam.max_time, device=dev)
    return scam, cam, stat, dyn, (stat_p, dyn_p, dyn_x)


def leaves(stat, dyn):
    ls = list(stat.leaf_tensors(False).values()) + list(dyn.leaf_tensors(True).values())
    return ls + list(dyn.rgbdecoder.parameters())


def cpu_baseline(stat_p, dyn_p, dyn_x, scam, width, height, reps):
    """C/OpenMP port of upstream's kernels (oracle/gsplat_cpu.c) on the activated splats of the same scene:
    rasterization() forward+backward, 10 channels (9 features + depth), same image size."""
    import numpy as np
    from oracle import gsplat_cpu as Cc
    from oracle import render_torch as R
    ctrl = R.hermite(dyn_x["control_xyz"], torch.tensor(scam.time), dyn_x["current_control_num"]) * 1e-2
    tfp = scam.time - dyn_x["trbf_center"]
    means = torch.cat([stat_p["xyz"], ctrl]).numpy()
    quats = torch.cat([stat_p["rotation"], dyn_p["rotation"] + tfp * dyn_x["omega"]]).numpy()
    scales = torch.exp(torch.cat([stat_p["scaling"], dyn_p["scaling"]])).numpy()
    opac = torch.sigmoid(torch.cat([stat_p["opacity"], dyn_p["opacity"]])).squeeze(-1).numpy()
    cols = torch.cat([torch.cat([stat_p["features_dc"], 0 * stat_p["features_t"]], 1),
                      torch.cat([dyn_p["features_dc"], tfp * dyn_p["features_t"]], 1)]).numpy()
    v = np.random.default_rng(0).standard_normal((1, height, width, 10)).astype(np.float32)
    args = (means, quats, scales, opac, cols, np.eye(4, dtype=np.float32)[None], scam.K.numpy()[None], width, height)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = Cc.rasterization_fwd_bwd(*args, backgrounds=np.zeros((1, 9), np.float32), render_mode="RGB+ED", v_render=v)
    dt = (time.perf_counter() - t0) / reps
    return {"value": 1.0 / dt, "unit": "renders/s", "cores": Cc.num_threads(), "kind": "port",
            "sample": f"{reps} x rasterization fwd+bwd of the full workload ({means.shape[0]} splats, "
                      f"{width}x{height}, I={int(r['flatten_ids'].shape[0])}) by oracle/gsplat_cpu.c (OpenMP)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--ns", type=int, default=200_000)
    ap.add_argument("--nd", type=int, default=100_000)
    ap.add_argument("--width", type=int, default=1352)
    ap.add_argument("--height", type=int, default=1014)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=2)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    # MOBGS_BENCH_SHARE_GPU=1 + MOBGS_BENCH_BACKEND=gloo: functional check of the N>1 code path on a ONE-GPU box
    # (all ranks on cuda:0, collectives through gloo); never used for reported numbers
    share = os.environ.get("MOBGS_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("MOBGS_BENCH_BACKEND", "nccl")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mobgs_amd.distributed import SubframeShard
    from mobgs_amd.gaussian_renderer import render

    scam, cam, stat, dyn, raw = build_scene(dev, args.ns, args.nd, args.width, args.height)
    bg = torch.zeros(9, device=dev)
    g = torch.Generator().manual_seed(100)
    v_render = torch.randn(3, args.height, args.width, generator=g).to(dev)
    v_depth = torch.randn(1, args.height, args.width, generator=g).to(dev)
    params = leaves(stat, dyn)
    shard = SubframeShard(world, rank)
    # one latent sub-frame per rank: exposure offsets linspace(-0.4, 0.4, world) (BLCE default, scene/blce.py)
    deltas = torch.linspace(-0.4, 0.4, world) if world > 1 else torch.zeros(1)
    delta = None if world == 1 else deltas[rank].to(dev)

    def step():
        for p in params:
            p.grad = None
        out = render(cam, stat, dyn, None, bg, delta_exposure=delta)
        pred = shard.mean_of_subframes(out["render"], world)  # all-reduce(SUM)/K + 1e-10 when world > 1
        # back-propagate fixed random cotangents: d(loss)/d(pred) = v_render, d(loss)/d(depth) = v_depth
        torch.autograd.backward([pred, out["depth"]], [v_render, v_depth])
        shard.all_reduce_gradients(params)
        return out

    # backward on the calling thread: handing each backward pass to autograd's device thread costs ~0.35 ms of
    # wake-up latency per step on this host (scripts/autograd_threads.py: 30 k splats 0.82 -> 0.47 ms/step; nothing
    # at 300 k, where the step is GPU-bound) -- the setting a training script on this stack would use
    torch.autograd.set_multithreading_enabled(False)
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    profiler.enable(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = profiler.summary()
    profiler.enable(False)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # workload statistics for the roofline: intersections I and pixels P of this rank's render
    from mobgs_amd import rendering
    I = rendering.last_stats.get("n_isects", 0)
    P = args.width * args.height
    n_vis = int((out["radii"] > 0).sum())
    ms_per_step = dt / args.steps * 1e3
    value = world * args.steps / dt  # every rank completes one render fwd+bwd per step

    result = {
        "metric": "train-step renders/sec (fwd+bwd, 1352x1014, 300k Gaussians)",
        "value": round(value, 3),
        "unit": "renders/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not share else "synthetic (FUNCTIONAL CHECK: ranks share one GPU, not a measurement)",
        "config": {"workload": "BASELINE config #2: seesaw-synth (SURVEY 8d seed 0), "
                               f"{args.ns} static + {args.nd} dynamic Gaussians, {args.width}x{args.height}, "
                               "render() lean mode fwd+bwd incl. spline prep, decoder, camera gradient",
                   "gaussians": args.ns + args.nd, "visible": n_vis, "intersections": I, "pixels": P,
                   "subframes_per_step": world, "parallelism": f"subframe-shard x{world}" if world > 1 else "single"},
    }
    if rank == 0:
        rb = prof.get("raster_bwd")
        if rb:
            # algorithmic bytes of raster_bwd per launch (DESIGN.md "Kernels"): per intersection 4 (id) + 64 (splat
            # record) gathered + 64 (gradient record) written; per pixel 40 (v_render) + 4 (v_alpha) + 4 (alpha)
            # + 4 (last_id) read
            alg_bytes = 132.0 * I + 52.0 * P
            achieved = alg_bytes / (rb["avg_ms"] * 1e-3) / 1e9
            traffic = valu_insts = None
            pmc = os.path.join(ROOT, "profiles", "r01_raster_bwd_pmc.json")
            if os.path.exists(pmc):
                try:
                    counters = json.load(open(pmc))
                    traffic = counters.get("hbm_bytes_per_launch")
                    valu_insts = counters.get("valu_wave_insts_per_launch")
                except Exception:  # noqa: BLE001
                    traffic = valu_insts = None
            result["roofline"] = {"kernel": "raster_bwd_kernel<10>", "bound": "hbm", "achieved": round(achieved, 2),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                                  "traffic": traffic, "avg_kernel_ms": round(rb["avg_ms"], 4),
                                  "algorithmic_bytes": alg_bytes, "calls": rb["calls"]}
            if valu_insts:
                # what actually bounds the kernel: VALU issue.  A wave64 VALU instruction holds
                # one of the 1024 SIMDs for 4 cycles; counted instructions (SQ_INSTS_VALU, committed PMC pass) x 4 /
                # 1024 / 2.4 GHz peak clock against the live kernel time (the chip sustains ~2.2 GHz here, so the
                # true occupancy of the issue slots is ~9 % higher than this figure)
                floor_ms = valu_insts * 4.0 / 1024.0 / 2.4e9 * 1e3
                result["roofline"]["valu_issue"] = {"wave_insts": valu_insts, "floor_ms_at_2.4GHz": round(floor_ms, 4),
                                                    "frac": round(floor_ms / rb["avg_ms"], 4)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(*raw, scam, args.width, args.height, args.cpu_reps)
            except Exception as exc:  # noqa: BLE001
                result["cpu_baseline"] = {"error": str(exc)}
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
