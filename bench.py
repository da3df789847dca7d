#!/usr/bin/env python
"""Benchmark of the MoBGS render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): train-step renders/sec (fwd+bwd, 1352x1014, 300k Gaussians) at 1/2/4/8 GPU.

N = 1   one step = one full `render()` forward + backward of BASELINE config #2: the synthetic "seesaw" scene of
        SURVEY.md section 8d (200k static + 100k dynamic Gaussians, 1352x1014, lean mode as eval.py:125), i.e. per-splat
        prep (Hermite spline, activations) -> projection -> tile lists + per-tile depth sort -> 10-channel compositing
        -> expected depth + colour decoder, and the backward pass to every Gaussian leaf, the decoder weights and the
        camera matrix.  `value` = renders/s.  The same line carries, as secondary objects, `deblur` (the K = 9
        blurry-view throughput on this one GPU: the N > 1 step below with world = 1), `dynamic_config3` (BASELINE
        config #3) and `get_flow` (the nine get_flow() calls train.py issues per view, :570-579).
N > 1   one step = one training iteration's blurry-view part (train.py:430-541) for a batch of 2 views (the
        reference's batch_size, arguments/stereo/default.py): per view 1 train-mode mid render + 8 latent renders
        through BLCE-warped cameras with exposure offsets, K = 9; the 18 (view, sub-frame) units are sharded
        round-robin over the ranks (mobgs_amd.distributed), the partial image sums are all-reduced (RCCL) into the
        blurry predictions, fixed cotangents are back-propagated (prediction on every rank, depth / mask terms on the
        rank that rendered the mid frame) and ONE in-place all-reduce of the persistent flat gradient buffer (all
        Gaussian leaves, decoder, BLCE parameters, mid-frame densification statistics) completes the step.
        `value` = 18 renders x steps / time = whole-job renders/s; total work per step is fixed: "scaling": "strong".

Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0 (see the task contract) with
  roofline     -- the dominant kernel (raster_bwd): algorithmic bytes / HIP-event time vs the 8 TB/s HBM peak, plus
                  pixel-splat pairs/s, fp32 FLOP/s against the 157.3 TFLOP/s vector peak
  cpu_baseline -- oracle/gsplat_cpu.c (OpenMP port of upstream's kernels) at the full workload and the north-star
                  "PyTorch-CPU render" (oracle/render_torch + gsplat_torch, config #1) timed on this host, N=1 only.
"""
from __future__ import annotations

import argparse
import gc
import hashlib
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

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_VECTOR_PEAK_TF = 157.3  # same guide: peak fp32 (vector)


def build_scene(dev, ns, nd, width, height, seed=0, sort=True):
    scam = SynthCamera().scaled(width, height) if (width, height) != (1352, 1014) else SynthCamera()
    stat_p = gaussian_cloud(ns, scam, seed)
    dyn_p = gaussian_cloud(nd, scam, seed + 1)
    dyn_x = dynamic_extras(dyn_p["xyz"], seed)
    torch.manual_seed(seed)
    dec = Sandwich(9, 3).to(dev)
    stat = GaussianParams(stat_p, None, dec, dev, requires_grad=True)
    dyn = GaussianParams(dyn_p, dyn_x, dec, dev, requires_grad=True)
    if sort and os.environ.get("MOBGS_BENCH_UNSORTED") != "1":
        # the model keeps its rows along a Morton curve (what a loader / densify.TrainableGaussians(keep_sorted) does after
        # loading and after every densification): the same Gaussians, rows permuted.  MOBGS_BENCH_UNSORTED=1: rows as
        # generated; the renderer then falls back to a cached enumeration order (+11 us of kernel time per step)
        stat.spatial_sort_()
        dyn.spatial_sort_()
    cam = PinholeCamera(width, height, scam.K, torch.eye(4), time=scam.time, max_time=scam.max_time, device=dev)
    return scam, cam, stat, dyn, (stat_p, dyn_p, dyn_x)


def leaves(stat, dyn):
    """The tensors an optimiser / flat gradient buffer holds: the leaves, or the fp32 masters of half-stored ones."""
    ls = list(stat.trainable_tensors(False).values()) + list(dyn.trainable_tensors(True).values())
    return ls + list(dyn.rgbdecoder.parameters())


def view_pose(i):
    """World-to-camera matrices of the batch's views: view 0 = identity, the others a few degrees / centimetres off."""
    import math
    w2c = torch.eye(4)
    if i:
        a, b = 0.03 * i, -0.02 * i
        Ry = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        Rx = torch.tensor([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
        w2c[:3, :3] = Ry @ Rx
        w2c[:3, 3] = torch.tensor([0.03 * i, -0.02 * i, 0.04 * i])
    return w2c


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


def cpu_torch_reference(width, height, ns=10_000):
    """The north star's "reference PyTorch-CPU render": BASELINE config #1 (10k static Gaussians, one sharp frame,
    FORWARD) through the restated reference glue (oracle/render_torch.render) over the pure-PyTorch restatement of
    gsplat (oracle/gsplat_torch) on the host cores.  One render (tens of seconds)."""
    from oracle import render_torch as R
    scam = SynthCamera().scaled(width, height) if (width, height) != (1352, 1014) else SynthCamera()
    stat_p = gaussian_cloud(ns, scam, 0)
    dyn_p = gaussian_cloud(8, scam, 1)  # the reference always renders both sets; 8 dynamic splats
    dyn_x = dynamic_extras(dyn_p["xyz"], 0)
    torch.manual_seed(0)
    dec = Sandwich(9, 3)
    cpu = torch.device("cpu")
    stat = GaussianParams(stat_p, None, dec, cpu, requires_grad=False)
    dyn = GaussianParams(dyn_p, dyn_x, dec, cpu, requires_grad=False)
    cam = PinholeCamera(width, height, scam.K, torch.eye(4), time=scam.time, max_time=scam.max_time, device=cpu)
    t0 = time.perf_counter()
    with torch.no_grad():
        out = R.render(cam, stat, dyn, torch.zeros(9))
    dt = time.perf_counter() - t0
    assert torch.isfinite(out["render"]).all()
    return {"value": 1.0 / dt, "unit": "renders/s (forward only)", "seconds": round(dt, 2),
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 x render() forward of BASELINE config #1 ({ns} static Gaussians, {width}x{height}) by "
                      "oracle/render_torch.py + oracle/gsplat_torch.py (PyTorch CPU)"}


def source_sha():
    """Identity of the compositor / binning sources a committed PMC summary was collected with: their code with
    comments and blank lines removed (a reworded comment does not make the counters stale)."""
    import re
    h = hashlib.sha256()
    for f in ("raster.hip", "raster_bwd_mfma.hip", "raster_shared.h", "decoder_shared.h", "isect.hip", "common.h"):
        with open(os.path.join(ROOT, "mobgs_amd", "csrc", f), "r", encoding="utf-8") as fh:
            text = fh.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        code = "\n".join(line.rstrip() for line in text.splitlines() if line.strip())
        h.update(code.encode())
    return h.hexdigest()[:16]


PMC_JSON = os.path.join(ROOT, "profiles", "r06_raster_bwd_pmc.json")


def raster_bwd_bytes(I, P):
    """Algorithmic bytes of the lean step's backward compositor per launch -> (total, per-pixel
    streaming part, what).  Per intersection 4 (id) + 64 (splat record) gathered + 64 (gradient record) written.  Per pixel,
    with the decoder's backward pass as the kernel's prologue (round 6, rendering.FUSE_DECODER_BWD): 40 (composited
    features) + 12 (cotangent of the decoded colour) + 4 (of the depth) + 4 (alpha) + 4 (last id) = 64 read; without it:
    40 (cotangent image) + 4 (alpha cotangent) + 4 + 4 = 52."""
    import mobgs_amd.rendering as _R
    per_px = 64.0 if (_R.FUSE_DECODER_BWD and _R.FUSE_DECODER) else 52.0
    return 132.0 * I + per_px * P, per_px * P, ("132 I + 64 P (decoder backward as the prologue)" if per_px == 64.0
                                                 else "132 I + 52 P")


def _rocprof_child(extra, tag, child_args):
    """Run `python bench.py <child_args>` under rocprofv3 with `extra` options in a scratch directory; -> that directory
    (None when rocprofv3 is not installed or the run failed).  Counter passes never carry tracing flags."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix=f"mobgs_{tag}_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [exe, *extra, "--output-format", "csv", "-d", out, "-o", tag, "--", sys.executable,
           os.path.join(ROOT, "bench.py"), *child_args]
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    except Exception:  # noqa: BLE001
        return None
    return out if r.returncode == 0 else None


def _find(root, suffix):
    for d, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(suffix):
                return os.path.join(d, f)
    return None


def kernel_breakdown(args, N, Ns, Nd, I, I_box, P):
    """Per-kernel time of the lean step from a rocprofv3 --kernel-trace --stats pass over a child run of this script
    (40 steps of the same workload; the kernels of one process cannot be timed one by one with events from Python:
    most of them are launched inside single C-ABI calls) -> {kernel: {avg_us, calls_per_step, bytes, GB/s, frac}} with
    the ALGORITHMIC bytes of DESIGN.md section 4 per launch.  Times under the profiler run ~2 % slow (clock)."""
    import csv
    steps = 40
    out = _rocprof_child(["--kernel-trace", "--stats"], "kb", ["--steps", str(steps), "--warmup", "10",
                         "--breakdown-child", "--ns", str(args.ns), "--nd", str(args.nd), "--width", str(args.width),
                         "--height", str(args.height)])
    f = _find(out, "kernel_stats.csv") if out else None
    if not f:
        return None
    algo = [  # (name as reported, substring of the kernel name, algorithmic bytes per launch)
        ("prep_fwd", "prep_fwd_kernel", 124.0 * Ns + 252.0 * Nd),
        ("prep_bwd", "prep_bwd_kernel", 124.0 * Ns + 252.0 * Nd + 80.0 * N),
        # round 5, lean render: the projection kernel builds the per-splat state itself (raw parameters in: 80 B per static,
        # 232 B per dynamic splat; state for the backward pass 44 B, projection outputs 32 B, compositor record 64 B and
        # bin record 48 B out) and the projection backward ends in the leaf gradients (state + cotangents in: 124 B per
        # splat; 80 B of gradients per static, 228 B per dynamic splat out) -- no prep_fwd / prep_bwd launch
        ("project_fwd(+prep, records, bin records)", "project_fwd_kernel", 80.0 * Ns + 232.0 * Nd + 188.0 * N),
        ("project_bwd(+prep_bwd)", "project_bwd_kernel", 124.0 * N + 80.0 * Ns + 228.0 * Nd),
        ("scan_lookback", "scan_lookback", 8.0 * N),
        # single-pass lists (round 5): per box intersection 4 B of the scan + its share of the 48-byte bin record (read
        # once per splat) + 4 B of keep_scan; per listed entry the 8-byte key written straight into its tile's segment
        ("bin", "bin_kernel", 8.0 * I_box + 48.0 * N + 8.0 * I),
        ("tile_finish", "tile_finish_kernel", None),
        ("tile_scan", "tile_scan_kernel", None),       # (two-pass path: the first frame of a workload / after an overflow)
        ("emit", "emit_kernel", 20.0 * I),
        ("tile_sort", "tile_sort", 12.0 * I),          # 8-byte key in, 4-byte id out
        # (+ the decoder epilogue of the lean render: 16 B per pixel of rgb + depth on top of the 48)
        ("raster_fwd(+decode)", "raster_fwd", 68.0 * I + 64.0 * P),
        ("raster_bwd(+decoder_bwd prologue)", "raster_bwd", raster_bwd_bytes(I, P)[0]),
        ("slot_reduce", "slot_reduce", 64.0 * I + 64.0 * N),
        ("slot_zero_fill", "FillFunctor", 64.0 * I),
        ("decoder_fwd", "decoder_fwd_kernel", 84.0 * P),
        ("decoder_bwd", "decoder_bwd_kernel", 132.0 * P),
        ("decoder_wgrad_reduce", "decoder_wgrad_reduce", None),
    ]
    rows = list(csv.DictReader(open(f)))
    total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    calls0 = steps + 10
    res, seen = {}, 0.0
    for name, pat, nbytes in algo:
        hit = [r for r in rows if pat in r["Name"]]
        if not hit:
            continue
        tot = sum(float(r["TotalDurationNs"]) for r in hit)
        calls = sum(int(r["Calls"]) for r in hit)
        seen += tot
        avg_us = tot / calls / 1e3
        e = {"avg_us": round(avg_us, 2), "launches_per_step": round(calls / calls0, 2)}
        if nbytes:
            gbs = nbytes / (avg_us * 1e-6) / 1e9
            e.update(bytes=nbytes, GBps=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))
        res[name] = e
    # kernels launched (about) once per step against the ones launched a few times per RUN (building the synthetic
    # scene, the first frame's two-pass lists, the enumeration order's sort: once per 2048 calls): the latter are not
    # a per-step cost and are listed as a total of their own instead of being divided by this short run's step count
    once = [r for r in rows if int(r["Calls"]) * 4 < calls0]
    once_ns = sum(float(r["TotalDurationNs"]) for r in once)
    once_seen = sum(float(r["TotalDurationNs"]) for r in once if any(p in r["Name"] for _, p, _ in algo))
    res["_all_kernels_us_per_step"] = round((total_ns - once_ns) / calls0 / 1e3, 1)
    res["_other_kernels_us_per_step"] = round((total_ns - once_ns - (seen - once_seen)) / calls0 / 1e3, 1)
    res["_per_run_kernels_us_total"] = round(once_ns / 1e3, 1)
    res["_all_kernels_incl_per_run_us_per_step"] = round(total_ns / calls0 / 1e3, 1)
    res["_source"] = f"rocprofv3 --kernel-trace --stats over a {steps}-step child run of this script (same workload)"
    import shutil
    shutil.rmtree(out, ignore_errors=True)
    return res


def collect_pmc(args, I, P):
    """bench.py --pmc: re-collect the counters behind roofline.traffic / roofline.valu for raster_bwd -- three separate
    rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU: MI355X_MICROARCH.md, "rocprofv3 PMC slots"; no
    tracing flags) over a short child run -- and store them with the hash of the kernel sources in
    profiles/r05_raster_bwd_pmc.json.  Correction rule (measured, DESIGN section 5: scripts/ubench/fetch_calib.hip):
    FETCH_SIZE counts a 16-byte-per-lane coalesced stream at 0.5x and 64-byte record gathers / stores at 1.0x."""
    import csv
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        out = _rocprof_child(["--pmc", ctr], ctr.lower(), ["--steps", "10", "--warmup", "2", "--breakdown-child",
                             "--ns", str(args.ns), "--nd", str(args.nd), "--width", str(args.width), "--height",
                             str(args.height)])
        f = _find(out, "counter_collection.csv") if out else None
        if not f:
            return {"error": f"no counter file for {ctr}"}
        tot, n = 0.0, 0
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == ctr and "raster_bwd" in row["Kernel_Name"]:
                tot += float(row["Counter_Value"])
                n += 1
        vals[ctr] = tot / max(1, n)
        import shutil
        shutil.rmtree(out, ignore_errors=True)
    fetch, write = vals["FETCH_SIZE"] * 1024.0, vals["WRITE_SIZE"] * 1024.0
    alg, stream, what = raster_bwd_bytes(I, P)
    d = {"source_sha": source_sha(), "kernel": "raster_bwd_kernel<10, false, DECB>", "algorithmic_bytes_formula": what,
         "source": "bench.py --pmc: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU (three separate passes over "
                   "bench.py --steps 10 --warmup 2 --breakdown-child)",
         "FETCH_SIZE_KiB_per_launch_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch_raw": vals["WRITE_SIZE"],
         "hbm_bytes_per_launch": fetch + 0.5 * stream + write, "hbm_bytes_per_launch_lower_estimate": fetch + write,
         "algorithmic_bytes": alg, "valu_wave_insts_per_launch": vals["SQ_INSTS_VALU"],
         "traffic_over_algorithmic": [round((fetch + write) / alg, 3), round((fetch + 0.5 * stream + write) / alg, 3)]}
    os.makedirs(os.path.dirname(PMC_JSON), exist_ok=True)
    json.dump(d, open(PMC_JSON, "w"), indent=1)
    return d


class DeblurWorkload:
    """The blurry-view part of one training iteration for a batch of views (train.py:430-541), see module docstring."""

    def __init__(self, dev, stat, dyn, scam, width, height, shard, n_views=2, n_sub=9, seed=100, batched=True):
        self.batched = batched
        from mobgs_amd.blce import blceKernel
        from mobgs_amd.distributed import FlatGradients
        self.dev, self.stat, self.dyn, self.shard, self.V, self.K = dev, stat, dyn, shard, n_views, n_sub
        g = torch.Generator().manual_seed(seed)
        self.cams = []
        for i in range(n_views):
            cam = PinholeCamera(width, height, scam.K, view_pose(i), time=scam.time, max_time=scam.max_time, device=dev)
            cam.uid = i
            cam.image = torch.rand(3, height, width, generator=g).to(dev)  # BLCE's blur statistic reads it once
            self.cams.append(cam)
        torch.manual_seed(seed)
        self.blce = blceKernel(num_views=n_views, num_warp=n_sub, iteration=10000).to(dev)
        self.bg = torch.zeros(9, device=dev)
        self.v_pred = torch.randn(n_views, 3, height, width, generator=g).to(dev)
        self.v_depth = torch.randn(1, height, width, generator=g).to(dev)
        self.v_alpha = torch.randn(1, height, width, generator=g).to(dev)
        self.params = leaves(stat, dyn) + list(self.blce.model.get_params())
        n = stat.get_xyz.shape[0] + dyn.get_xyz.shape[0]
        self.bucket = FlatGradients(self.params, extra={f"view{v}": 3 * n for v in range(n_views)})
        # N > 1: one gradient message PER VIEW (its densification statistics ride in it), all-reduced on the communication
        # stream while the next view back-propagates (SubframeShard.backward_by_view)
        self.view_buckets = ([FlatGradients(self.params, extra={f"view{v}": 3 * n}) for v in range(n_views)]
                             if shard.collective else None)
        self.mids = {}

    def step(self):
        from mobgs_amd.deblur import render_blurry_batch
        from mobgs_amd.ops import LeafGradSink
        self.bucket.zero()
        # N > 1: units dealt by cost (the two train-mode mid frames weigh 2.3 latent renders), one asynchronous image
        # all-reduce per view behind the next view's renders
        multi = self.shard.collective
        pred, mids = render_blurry_batch(self.cams, self.stat, self.dyn, self.bg, self.shard, blce=self.blce,
                                         n_sub=self.K, weighted=multi, overlap=multi, batched_latent=self.batched,
                                         as_list=multi)
        if multi:
            def view_backward(v):
                outs, cots = [pred[v]], [self.v_pred[v]]
                if v in mids:
                    for key in ("s_render", "s_depth", "d_alpha", "d_depth", "s_alpha"):
                        mids[v][key]
                    outs += [mids[v]["depth"], mids[v]["d_alpha"]]
                    cots += [self.v_depth, self.v_alpha]
                live = [(o, c) for o, c in zip(outs, cots) if o.requires_grad]
                if live:
                    with LeafGradSink(self.stat, self.dyn, extra=self.blce.model.get_params()):
                        torch.autograd.backward([o for o, _ in live], [c for _, c in live])

            def after_view(v):
                if v in mids:
                    self.shard.put_densification_stats(self.view_buckets[v], f"view{v}", mids[v]["viewspace_points"].grad,
                                                       mids[v]["radii"])
            self.shard.backward_by_view(self.view_buckets, view_backward, after_view)
            self.mids = mids
            return pred
        outs, cots = [pred], [self.v_pred]
        for v, pkg in mids.items():  # depth / mask terms live on the rank that rendered the mid frame
            for key in ("s_render", "s_depth", "d_alpha", "d_depth", "s_alpha"):
                pkg[key]             # train.py:445-464 reads these five auxiliary images of the mid render every iteration
            outs += [pkg["depth"], pkg["d_alpha"]]
            cots += [self.v_depth, self.v_alpha]
        if any(o.requires_grad for o in outs):
            with LeafGradSink(self.stat, self.dyn, extra=self.blce.model.get_params()):
                torch.autograd.backward([o for o in outs if o.requires_grad],
                                        [c for o, c in zip(outs, cots) if o.requires_grad])
        for v, pkg in mids.items():
            self.shard.put_densification_stats(self.bucket, f"view{v}", pkg["viewspace_points"].grad, pkg["radii"])
        self.shard.all_reduce_gradients(self.bucket)
        self.mids = mids
        return pred


    def step_unchanged(self):
        """The same blurry views the way train.py:441-541 asks for them after nothing but the import swap: one render()
        per latent sub-frame, every one with get_static = get_dynamic = True (train.py:441, :512), torch.mean of the stack,
        a plain backward into ordinary .grad tensors -- no render_many, no LeafGradSink, no flat gradient buffer."""
        for p in self.params:
            p.grad = None
        preds, outs, cots = [], [], []
        for cam in self.cams:
            pkg = render_fn(cam, self.stat, self.dyn, None, self.bg, get_static=True, get_dynamic=True)
            for key in ("s_render", "s_depth", "d_alpha", "d_depth", "s_alpha"):
                pkg[key]
            warped_cams, exposure_time = self.blce.get_warped_cams(cam, None, None)
            half = len(warped_cams) // 2
            rendered = [pkg["render"] if k == half else
                        render_fn(wc, self.stat, self.dyn, None, self.bg, get_static=True, get_dynamic=True,
                                  delta_exposure=exposure_time[k])["render"] for k, wc in enumerate(warped_cams)]
            preds.append(torch.mean(torch.stack(rendered, dim=0), dim=0) + 1e-10)
            outs += [pkg["depth"], pkg["d_alpha"]]
            cots += [self.v_depth, self.v_alpha]
        torch.autograd.backward([torch.stack(preds)] + outs, [self.v_pred] + cots)
        return preds


def render_fn(*a, **kw):
    from mobgs_amd.gaussian_renderer import render
    return render(*a, **kw)


class FlowWorkload:
    """The nine get_flow() calls train.py issues per view and iteration (:570-579, exposure offsets (k - 4) / 4),
    forward + backward, through get_flow_many; `zero` = the cotangents of lambda_flow_loss = 0 (arguments/stereo/
    seesaw.py: the flow loss is formed and back-propagated with weight 0)."""

    def __init__(self, dev, stat, dyn, cam, width, height, zero, seed=100):
        g = torch.Generator().manual_seed(seed)
        mk = (lambda *s: torch.zeros(*s, device=dev)) if zero else (lambda *s: torch.randn(*s, generator=g).to(dev))
        self.v2, self.v3, self.v1 = mk(1, height, width, 2), mk(3, height, width), mk(1, height, width)
        self.stat, self.dyn, self.cam = stat, dyn, cam
        self.bg = torch.zeros(9, device=dev)
        self.deltas = [(k - 4) / 4.0 for k in range(9)]
        self.params = leaves(stat, dyn)

    def step(self):
        from mobgs_amd.gaussian_renderer import get_flow_many
        from mobgs_amd.ops import LeafGradSink
        for p in self.params:
            p.grad = None
        outs = get_flow_many(self.cam, self.stat, self.dyn, None, self.bg, self.deltas)
        with LeafGradSink(self.stat, self.dyn):
            torch.autograd.backward([t for o in outs for t in o], [self.v2, self.v2, self.v3, self.v1] * 9)


class DynamicWorkload:
    """BASELINE config #3 (secondary object of the N = 1 line): `deform_network` (HexPlane [64,64,64,12] x [1,2,4] +
    MLP heads, arguments/stereo/seesaw.py) moves the dynamic Gaussians' position / scale / rotation, the result is
    rasterised together with the static set (RGB+ED, 9 channels) and fixed cotangents are back-propagated to the
    deformation network's planes and weights and to the Gaussian inputs."""

    def __init__(self, dev, raw, scam, width, height, seed=0):
        from mobgs_amd.deformation import SeesawArgs, deform_network
        stat_p, dyn_p, _ = raw
        torch.manual_seed(seed)
        self.net = deform_network(SeesawArgs()).to(dev)
        lo, hi = dyn_p["xyz"].min(0).values, dyn_p["xyz"].max(0).values
        self.net.deformation_net.set_aabb(hi.tolist(), lo.tolist())
        with torch.no_grad():
            for pl in self.net.deformation_net.grid.planes():
                pl.uniform_(0.5, 1.0)
        T = lambda t: t.to(dev)  # noqa: E731
        self.s = {"means": T(stat_p["xyz"]), "scales": T(torch.exp(stat_p["scaling"])), "quats": T(stat_p["rotation"])}
        self.leaves = [T(dyn_p[k]).requires_grad_(True) for k in ("xyz", "scaling", "rotation")]
        self.opac = T(torch.sigmoid(torch.cat([stat_p["opacity"], dyn_p["opacity"]])).squeeze(-1))
        self.cols = T(torch.cat([torch.cat([stat_p["features_dc"], stat_p["features_t"]], 1),
                                 torch.cat([dyn_p["features_dc"], dyn_p["features_t"]], 1)]))
        self.times = torch.full((dyn_p["xyz"].shape[0], 1), scam.time, device=dev)
        self.K, self.vm = T(scam.K)[None], torch.eye(4, device=dev)[None]
        self.W, self.H = width, height
        g = torch.Generator().manual_seed(seed + 5)
        self.v = torch.randn(1, height, width, 10, generator=g).to(dev)
        self.bg = torch.zeros(1, 9, device=dev)
        self.params = list(self.net.parameters())

    def step(self):
        from mobgs_amd.rendering import rasterization
        for p in self.params + self.leaves:
            p.grad = None
        d_pts, d_scl, d_rot = self.net(*self.leaves, self.times)
        img, _, _ = rasterization(torch.cat([self.s["means"], d_pts]), torch.cat([self.s["quats"], d_rot]),
                                  torch.cat([self.s["scales"], torch.exp(d_scl)]), self.opac, self.cols, self.vm, self.K,
                                  self.W, self.H, packed=False, backgrounds=self.bg, render_mode="RGB+ED")
        torch.autograd.backward([img], [self.v])


def scaling_diagnostics(wl, shard, args, dev, world, dist, steps=3):
    """What makes an N > 1 line explain itself, measured on `steps` extra iterations OUTSIDE the timed
    region with MOBGS_COMM_LOG=1 (the consumer-side waits of mobgs_amd.distributed are bracketed by HIP events):
      per rank: busy_ms (its compute stream's iteration time minus what it stalled on exchanges),
                image_exchange_exposed_ms / grad_exchange_exposed_ms (stalls on the per-view image sums / gradient messages);
      the partition's own ceiling (planned loads: total / (ranks x largest load));
      and an assertion that the all-reduced prediction is BIT-IDENTICAL on every rank (one 8-byte checksum per rank)."""
    from mobgs_amd import distributed as D
    old = os.environ.get("MOBGS_COMM_LOG")
    os.environ["MOBGS_COMM_LOG"] = "1"
    rows = []
    pred = None
    try:
        for _ in range(steps):
            del D.wait_log[:]
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            pred = wl.step()
            t1.record()
            torch.cuda.synchronize()
            img = sum(b.elapsed_time(a) for tag, b, a in D.wait_log if tag == "image")
            grad = sum(b.elapsed_time(a) for tag, b, a in D.wait_log if tag.startswith("grad"))
            total = t0.elapsed_time(t1)
            rows.append((total - img - grad, img, grad, total))
    finally:
        if old is None:
            os.environ.pop("MOBGS_COMM_LOG", None)
        else:
            os.environ["MOBGS_COMM_LOG"] = old
        del D.wait_log[:]
    cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")  # (gloo: gather on the host)
    mine = torch.tensor([sum(r[k] for r in rows) / len(rows) for k in range(4)], device=cdev, dtype=torch.float64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    # the prediction every rank holds after the image exchange: a checksum of its bits
    stack = torch.stack([p.detach() for p in pred]) if isinstance(pred, (list, tuple)) else pred.detach()
    chk = stack.contiguous().view(torch.int32).to(torch.int64).sum().reshape(1).to(cdev)
    chks = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(chks, chk)
    same = all(int(c.item()) == int(chks[0].item()) for c in chks)
    assert same, f"the all-reduced prediction differs between ranks: checksums {[int(c.item()) for c in chks]}"
    plan = shard.iteration_plan(args.views, 9, False)
    loads = plan["loads"]
    return {"per_rank": [{"rank": r, "busy_ms": round(float(e[0]), 3), "image_exchange_exposed_ms": round(float(e[1]), 3),
                          "grad_exchange_exposed_ms": round(float(e[2]), 3), "iteration_ms": round(float(e[3]), 3)}
                         for r, e in enumerate(every)],
            "planned_loads_latent_render_units": [round(x, 2) for x in loads],
            "planned_efficiency_ceiling": round(sum(loads) / (world * max(loads)), 4),
            "prediction_bit_identical_across_ranks": same,
            "note": "busy_ms = iteration time on the rank's compute stream minus its stalls on exchanges (HIP events around "
                    "every consumer-side wait); on a gloo group (functional runs) the exchanges are synchronous host copies "
                    "and show up as busy time"}


def freeze_python_heap():
    gc.collect()
    gc.freeze()


def timed(step, steps, warmup, world, dist, freeze=True):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; per-step HIP events on the current
    stream give the median next to the wall-clock mean."""
    # Host setting a training script on this stack would use (like the autograd threading switch in main()): a
    # generation-2 pass of Python's cyclic collector over the ~270 k objects a torch process holds takes ~66 ms on this
    # host (scripts/find_hiccup.py) -- during which nothing is enqueued and the device runs dry; it fires every few
    # dozen iterations, in the middle of one.  Everything alive after set-up moves to the permanent generation: later
    # collections only look at what the iterations themselves allocate.  (BEFORE the warm-up steps, not between them
    # and the timed ones: the collection itself idles the device for ~0.1 s and the clocks drop.)
    # freeze=False: the caller has done it during its own (longer) set-up steps.
    if freeze:
        if warmup > 0:   # the first of the W warm-up steps, then the collection, then the other W - 1
            step()
            torch.cuda.synchronize()
        freeze_python_heap()
        for _ in range(max(warmup - 1, 0)):
            step()
    else:
        for _ in range(warmup):
            step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    before = stall_counters()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    last_timed.clear()
    last_timed.update(event_ms_min=round(per[0], 3), event_ms_median=round(per[len(per) // 2], 3),
                      event_ms_max=round(per[-1], 3), **stall_counters(before))
    return dt, per[len(per) // 2]


last_timed = {}   # per-step HIP-event extremes of the most recent timed() region + what could have stalled it


def stall_counters(since=None):
    """Everything known to stall a step, as counts: tile-list rebuilds / key-segment overflows (rendering), device
    allocations and retries of the caching allocator (a hipMalloc synchronises the device), generation-2 collections.
    since: an earlier snapshot -> the differences (say what happened INSIDE the timed region)."""
    from mobgs_amd import rendering as R
    ms = torch.cuda.memory_stats() if torch.cuda.is_available() else {}
    now = dict(list_rebuilds=R.list_rebuilds[0], seg_overflows=R.seg_overflows[0],
               device_allocs=int(ms.get("num_device_alloc", 0)), alloc_retries=int(ms.get("num_alloc_retries", 0)),
               gc_gen2=gc.get_stats()[2]["collections"])
    return now if since is None else {k: now[k] - since[k] for k in now}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ns", type=int, default=200_000)
    ap.add_argument("--nd", type=int, default=100_000)
    ap.add_argument("--width", type=int, default=1352)
    ap.add_argument("--height", type=int, default=1014)
    ap.add_argument("--views", type=int, default=2, help="views per training iteration in the deblur step")
    ap.add_argument("--deblur-steps", type=int, default=10, help="N=1: steps of the secondary deblur leg (0: skip)")
    ap.add_argument("--dynamic-steps", type=int, default=20, help="N=1: steps of the secondary config #3 leg (0: skip)")
    ap.add_argument("--flow-steps", type=int, default=4, help="N=1: steps of the secondary get_flow leg (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-torch", action="store_true", help="skip the PyTorch-CPU config #1 render (tens of s)")
    ap.add_argument("--cpu-reps", type=int, default=2)
    ap.add_argument("--prewarm", type=int, default=40, help="untimed set-up steps before the W warm-up steps (arena sizing)")
    ap.add_argument("--train-steps", type=int, default=10, help="N=1: whole training iterations timed (0: skip)")
    ap.add_argument("--small-steps", type=int, default=20,
                    help="N=1: whole iterations at the reference's own 512x288 / 30 k operating point, eager and graphed (0: skip)")
    ap.add_argument("--repeat-steps", type=int, default=100,
                    help="N=1: further lean steps after the K timed ones; their HIP-event median is reported as `repeat`")
    ap.add_argument("--no-kernel-breakdown", action="store_true",
                    help="skip the rocprofv3 child run behind roofline.streaming")
    ap.add_argument("--pmc", action="store_true",
                    help="re-collect roofline.traffic / roofline.valu (three rocprofv3 --pmc child runs) first")
    ap.add_argument("--breakdown-child", action="store_true", help=argparse.SUPPRESS)
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
    last = {}
    # the headline workload includes the camera-pose gradient (config.workload says so; eval.py's test-time pose
    # optimisation needs it): its camera ASKS for it -- since round 5 the projection backward skips the pose gradient when
    # nobody does (train.py never optimises the pose), and the headline must not get faster by doing less
    cam_lean = PinholeCamera(args.width, args.height, scam.K, torch.eye(4), time=scam.time, max_time=scam.max_time,
                             device=dev)
    cam_lean.world_view_transform.requires_grad_(True)

    def lean_step():
        for p in params:
            p.grad = None
        cam_lean.world_view_transform.grad = None
        out = render(cam_lean, stat, dyn, None, bg)
        # back-propagate fixed random cotangents: d(loss)/d(pred) = v_render, d(loss)/d(depth) = v_depth
        torch.autograd.backward([out["render"], out["depth"]], [v_render, v_depth])
        last["out"] = out

    # backward on the calling thread: handing each backward pass to autograd's device thread costs ~0.35 ms of
    # wake-up latency per step on this host (scripts/autograd_threads.py: 30 k splats 0.82 -> 0.47 ms/step; nothing
    # at 300 k, where the step is GPU-bound) -- the setting a training script on this stack would use
    torch.autograd.set_multithreading_enabled(False)

    P = args.width * args.height
    n_units = args.views * 9
    deblur = dynamic = flows = train_it = repeat = small_scene = None
    if args.breakdown_child:  # the profiled child of kernel_breakdown() / collect_pmc(): lean steps only, no output
        for _ in range(args.steps + args.warmup):
            lean_step()
        torch.cuda.synchronize()
        return
    if world == 1:
        # set-up, not measurement: the first frames of a workload size the speculative arenas (one synchronous rebuild)
        # and bring the clocks up; with a small --warmup (the driver's invocation) the K timed steps would otherwise
        # include that transient (20 steps after 5 warm-ups: 922 renders/s against 952 in steady state)
        lean_step()
        torch.cuda.synchronize()
        freeze_python_heap()            # (see timed(): here, so that the set-up steps below re-heat the device after it)
        for _ in range(args.prewarm):
            lean_step()
        profiler.enable(True)
        dt, med_ms = timed(lean_step, args.steps, args.warmup, world, dist, freeze=False)
        prof = profiler.summary()
        profiler.enable(False)
        if args.repeat_steps > 0:  # the driver's K may be small: a longer run next to it (not `value`)
            rdt, rmed = timed(lean_step, args.repeat_steps, 0, world, dist, freeze=False)
            repeat = {"steps": args.repeat_steps, "ms_per_step": round(rdt / args.repeat_steps * 1e3, 4),
                      "event_median_ms_per_step": round(rmed, 4),
                      "renders_per_s": round(args.repeat_steps / rdt, 2)}

            def forward_only():  # eval.py / render.py: render() under no_grad (same scene, same camera)
                with torch.no_grad():
                    last["eval"] = render(cam, stat, dyn, None, bg)["render"]
            fdt, fmed = timed(forward_only, args.repeat_steps, 10, world, dist, freeze=False)
            repeat["forward_only_no_grad"] = {"ms_per_render": round(fdt / args.repeat_steps * 1e3, 4),
                                              "event_median_ms": round(fmed, 4),
                                              "renders_per_s": round(args.repeat_steps / fdt, 1),
                                              "what": "render() under torch.no_grad() (the eval.py / render.py use)"}
        from mobgs_amd import rendering
        I = rendering.last_stats.get("n_isects", 0)  # of the PRIMARY workload (the secondary legs render too)
        I_box = rendering.last_stats.get("n_box", 0) or I
        n_vis = int((last["out"]["radii"] > 0).sum())
        renders = args.steps
        workload = ("BASELINE config #2: seesaw-synth (SURVEY 8d seed 0), "
                    f"{args.ns} static + {args.nd} dynamic Gaussians, {args.width}x{args.height}, "
                    "render() lean mode fwd+bwd incl. spline prep, decoder, camera gradient")
        if args.deblur_steps > 0:
            wl = DeblurWorkload(dev, stat, dyn, scam, args.width, args.height, shard, args.views)
            ddt, dmed = timed(wl.step, args.deblur_steps, 10, world, dist)  # (the C = 8 batches size their arenas first)
            deblur = {"blurry_views_per_s": round(args.views * args.deblur_steps / ddt, 3),
                      "renders_per_s": round(n_units * args.deblur_steps / ddt, 2),
                      "ms_per_iteration": round(ddt / args.deblur_steps * 1e3, 3),
                      "event_median_ms_per_iteration": round(dmed, 3), "views_per_iteration": args.views,
                      "subframes_per_view": 9, "steps": args.deblur_steps,
                      "what": "train.py:430-541 per iteration: per view 1 train-mode mid render + 8 latent renders "
                              "(BLCE cameras + exposure offsets from the fused BLCE kernels), mean, backward, flat gradient buffer"}
            # ... and what the UNCHANGED caller gets (north_star: train.py drops in unchanged): the same views through one
            # render() call per sub-frame, all in train mode, plain autograd -- no opt-in entry point
            # ... on rows AS GENERATED: a reference GaussianModel knows nothing of spatial_sort_() (VERDICT r5 weak #4;
            # the one-line opt-in is INTEGRATION.md section B1, and it is not part of the word "unchanged")
            deblur["rows"] = "Morton-sorted at build (GaussianParams.spatial_sort_())"
            _, _, stat_u, dyn_u, _ = build_scene(dev, args.ns, args.nd, args.width, args.height, sort=False)
            wl_u = DeblurWorkload(dev, stat_u, dyn_u, scam, args.width, args.height, shard, args.views)
            udt, umed = timed(wl_u.step_unchanged, args.deblur_steps, 3, world, dist)
            del wl_u, stat_u, dyn_u
            deblur["unchanged_caller"] = {
                "ms_per_iteration": round(udt / args.deblur_steps * 1e3, 3), "event_median_ms_per_iteration": round(umed, 3),
                "renders_per_s": round(n_units * args.deblur_steps / udt, 2),
                "rows": "as generated (unsorted)",
                "what": "the same two blurry views as train.py:441-541 issues them after the import swap of INTEGRATION.md "
                        "section 1 alone: 9 render(get_static=True, get_dynamic=True) calls per view, torch.mean, backward "
                        "into ordinary .grad tensors (no render_many / LeafGradSink / FlatGradients)"}
        if args.dynamic_steps > 0:
            dw = DynamicWorkload(dev, raw, scam, args.width, args.height)
            xdt, xmed = timed(dw.step, args.dynamic_steps, 3, world, dist)
            dynamic = {"steps_per_s": round(args.dynamic_steps / xdt, 2),
                       "ms_per_step": round(xdt / args.dynamic_steps * 1e3, 4), "event_median_ms_per_step": round(xmed, 4),
                       "what": "BASELINE config #3: deform_network (seesaw HexPlane + MLP heads, HIP fwd + bwd) on the "
                               f"{args.nd} dynamic Gaussians -> rasterization of all {args.ns + args.nd} (RGB+ED) -> "
                               "backward to the network's planes / weights and the Gaussian inputs"}
        if args.flow_steps > 0:
            flows = {"what": "train.py:570-579: the nine get_flow() calls of one view (9 features + 2 flow channels + "
                             "dynamic coverage per call, mid-exposure state shared through get_flow_many), fwd + bwd"}
            for name, zero in (("ms_per_view", False), ("ms_per_view_zero_weight", True)):
                fw = FlowWorkload(dev, stat, dyn, cam, args.width, args.height, zero)
                fdt, _ = timed(fw.step, args.flow_steps, 2, world, dist)
                flows[name] = round(fdt / args.flow_steps * 1e3, 3)
            flows["zero_weight_note"] = "cotangents exactly zero: lambda_flow_loss = 0 (arguments/stereo/seesaw.py)"
            # what the same nine calls cost when the caller does not ask for gradients at all -- the one-line change
            # INTEGRATION.md suggests for configurations with lambda_flow_loss = 0 (autograd cannot know the weight is 0)
            from mobgs_amd.gaussian_renderer import get_flow_many
            fw = FlowWorkload(dev, stat, dyn, cam, args.width, args.height, True)

            def fwd_only():
                with torch.no_grad():
                    get_flow_many(fw.cam, fw.stat, fw.dyn, None, fw.bg, fw.deltas)
            fdt, _ = timed(fwd_only, args.flow_steps, 2, world, dist)
            flows["ms_per_view_no_grad"] = round(fdt / args.flow_steps * 1e3, 3)
        if args.train_steps > 0:
            sys.path.insert(0, os.path.join(ROOT, "examples"))
            import train_deblur_synth as TD
            tr = TD.DeblurTrainer(str(dev), args.ns, args.nd, args.width, args.height, args.views, shard=shard,
                                  iters=10000)
            # 4 untimed iterations: the arenas of the trainer's workloads (1-camera mid render, 8-camera batches, the
            # get_flow groups) reach their sizes -- round 5 timed 3 iterations after 2 and caught a 1.4-GB hipMalloc inside
            # them (61.7 ms mean against a 48.3 ms median); `stalls` says what happened inside THIS timed region
            tdt, tmed = timed(tr.iteration, args.train_steps, 4, world, dist)
            t_stats = dict(last_timed)
            tr.lambda_flow = 0.0   # the shipped configs (arguments/stereo/seesaw.py): calls made, no flow term in the graph
            zdt, _ = timed(tr.iteration, args.train_steps, 2, world, dist)
            train_it = {"ms_per_iteration": round(tdt / args.train_steps * 1e3, 2),
                        "ms_per_iteration_lambda_flow_loss_0": round(zdt / args.train_steps * 1e3, 2),
                        "event_median_ms_per_iteration": round(tmed, 2),
                        "event_max_ms_per_iteration": t_stats.get("event_ms_max"),
                        "stalls_inside_timed_region": {k: v for k, v in t_stats.items() if not k.startswith("event_")},
                        "rows": "as generated (unsorted); TrainableGaussians(keep_sorted) orders them at the first densification",
                        "iterations_per_s": round(args.train_steps / tdt, 3),
                        "steps": args.train_steps, "views_per_iteration": args.views,
                        "what": "ONE whole training iteration (train.py:430-807) at the headline size: per view K = 9 "
                                "latent renders through BLCE cameras (mid frame in train mode) + the 9 get_flow() calls, "
                                "fused L1 + D-SSIM on the blurry prediction, depth / mask / normal (get_normals) terms on "
                                "the mid render, the flow-consistency term of train.py:651-671 (two grid_sample warps + masked L1, fused), backward into the flat gradient buffer, "
                                "densification statistics, Adam on both Gaussian sets + decoder + BLCE "
                                "(examples/train_deblur_synth.py DeblurTrainer.iteration)"}
            tr.lambda_flow = 1e-2
            udt, umed = timed(tr.iteration_unchanged, args.train_steps, 1, world, dist)
            tr.lambda_flow = 0.0
            uzdt, _ = timed(tr.iteration_unchanged, args.train_steps, 1, world, dist)
            train_it["unchanged_caller"] = {
                "ms_per_iteration": round(udt / args.train_steps * 1e3, 2),
                "ms_per_iteration_lambda_flow_loss_0": round(uzdt / args.train_steps * 1e3, 2),
                "event_median_ms_per_iteration": round(umed, 2),
                "what": "the same iteration as train.py:430-807 writes it, after the import swap alone: per view 9 render() "
                        "(train mode) + 9 get_flow() calls, l1_loss + ssim as two calls (the fused kernels, reached through the "
                        "loss_utils shim of INTEGRATION.md), the flow-consistency term as two F.grid_sample + "
                        "two masked l1_loss (torch), loss.backward() into ordinary .grad tensors, three optimizer.step() calls "
                        "(optim.FusedAdam -- the torch.optim.Adam subclass densify.TrainableGaussians.training_setup and "
                        "blceKernel build: one launch each; a reference GaussianModel kept as is steps torch's Adam) "
                        "(DeblurTrainer.iteration_unchanged)"}
            # the same iteration with its forward + loss + backward recorded ONCE as a HIP graph (graphed.GraphedCallable;
            # the one-launch Adam step outside): at this size the iteration is device-bound -- what a graph removes is gaps
            try:
                from mobgs_amd.graphed import GraphedCallable
                tr.lambda_flow = 1e-2
                fb = GraphedCallable(tr.forward_backward, warmup=0)
                fb()

                def graphed_iteration():
                    fb()
                    tr.optimizer_step()
                gdt, gmed = timed(graphed_iteration, args.train_steps, 2, world, dist)
                train_it["graphed"] = {"ms_per_iteration": round(gdt / args.train_steps * 1e3, 2),
                                       "event_median_ms_per_iteration": round(gmed, 2), "arenas_fitted": bool(fb.check()),
                                       "what": "DeblurTrainer.forward_backward as one HIP graph + fused_adam_step"}
                del fb
            except Exception as exc:  # noqa: BLE001
                train_it["graphed"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            del tr
        if args.small_steps > 0:
            # The reference's OWN operating point (scene/dataset_readers.py:1448-1460, arguments/stereo/seesaw.py:13-14):
            # 512x288 images, ~30 k Gaussians at the start of training.  An iteration is ~1400 launches of a few microseconds
            # and bound by the host; recorded as a HIP graph it is bound by the device again.
            sys.path.insert(0, os.path.join(ROOT, "examples"))
            import train_deblur_synth as TD
            from mobgs_amd.graphed import GraphedCallable
            small = {"what": "ONE whole training iteration (as train_iteration) at 512x288 / 20 k + 10 k Gaussians, 2 views: "
                             "eager, and with DeblurTrainer.forward_backward replayed as one HIP graph (Adam outside)",
                     "steps": args.small_steps}
            for lam, tag in ((1e-2, ""), (0.0, "_lambda_flow_loss_0")):
                try:
                    trs = TD.DeblurTrainer(str(dev), 20_000, 10_000, 512, 288, 2, iters=10000, lambda_flow=lam)
                    edt, _ = timed(trs.iteration, args.small_steps, 4, world, dist)
                    fbs = GraphedCallable(trs.forward_backward, warmup=0)
                    fbs()

                    def graphed_small():
                        fbs()
                        trs.optimizer_step()
                    sdt, _ = timed(graphed_small, args.small_steps, 3, world, dist)
                    small["eager_ms_per_iteration" + tag] = round(edt / args.small_steps * 1e3, 3)
                    small["graphed_ms_per_iteration" + tag] = round(sdt / args.small_steps * 1e3, 3)
                    small["arenas_fitted" + tag] = bool(fbs.check())
                    del fbs, trs
                except Exception as exc:  # noqa: BLE001
                    small["error" + tag] = f"{type(exc).__name__}: {exc}"[:300]
            small_scene = small
    else:
        wl = DeblurWorkload(dev, stat, dyn, scam, args.width, args.height, shard, args.views)
        wl.step()
        torch.cuda.synchronize()
        freeze_python_heap()
        for _ in range(min(args.prewarm, 8)):   # set-up: arena sizing of this rank's batches (see the N = 1 branch)
            wl.step()
        profiler.enable(True)
        dt, med_ms = timed(wl.step, args.steps, args.warmup, world, dist, freeze=False)
        prof = profiler.summary()
        profiler.enable(False)
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        renders = n_units * args.steps
        # the anchor of the scaling curve, measured in THIS run: the same 18-unit iteration with all units on one GPU (no
        # collective), every rank on its own device, outside the timed region; efficiency(N) = value / (N x scale_anchor)
        awl = DeblurWorkload(dev, stat, dyn, scam, args.width, args.height, SubframeShard(1, 0), args.views)
        adt, _ = timed(awl.step, max(args.steps // 2, 3), 3, world, dist, freeze=False)
        t = torch.tensor([adt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        anchor = n_units * max(args.steps // 2, 3) / float(t.item())
        del awl
        scale_diag = scaling_diagnostics(wl, shard, args, dev, world, dist)
        last["out"] = next(iter(wl.mids.values())) if wl.mids else render(cam, stat, dyn, None, bg)
        from mobgs_amd import rendering
        I = rendering.last_stats.get("n_isects", 0)
        I_box = rendering.last_stats.get("n_box", 0) or I
        n_vis = int((last["out"]["radii"] > 0).sum())
        workload = (f"K=9 deblur iteration (train.py:430-541): {args.views} blurry views x (1 train-mode mid render + 8 "
                    f"latent renders, BLCE cameras + exposure offsets), seesaw-synth {args.ns} static + {args.nd} "
                    f"dynamic Gaussians, {args.width}x{args.height}, fwd+bwd, {n_units} (view, sub-frame) units "
                    f"dealt by cost over {world} ranks (planned loads {[round(x, 1) for x in shard.iteration_plan(args.views, 9, False)['loads']]} "
                    f"latent-render units), per-view asynchronous image all-reduce; value counts all {n_units} renders of a step")

    # workload statistics for the roofline: intersections I and pixels P of this rank's render (taken above)
    ms_per_step = dt / args.steps * 1e3
    value = renders / dt

    result = {
        "metric": "train-step renders/sec (fwd+bwd, 1352x1014, 300k Gaussians)",
        "value": round(value, 3),
        "unit": "renders/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "prewarm_steps": args.prewarm if world == 1 else min(args.prewarm, 8),
        "ms_per_step": round(ms_per_step, 4),
        "event_median_ms_per_step": round(med_ms, 4),
        "higher_is_better": True,
        "scaling": "weak" if world == 1 else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not share else "synthetic (FUNCTIONAL CHECK: ranks share one GPU, not a measurement)",
        "config": {"workload": workload, "gaussians": args.ns + args.nd, "visible": n_vis, "intersections": I,
                   "pixels": P, "subframes_per_step": 1 if world == 1 else n_units,
                   "renders_per_step": 1 if world == 1 else n_units,
                   "parallelism": f"subframe-shard x{world}" if world > 1 else "single",
                   "row_order": ("as generated (MOBGS_BENCH_UNSORTED=1): cached enumeration order in the binning kernel"
                                 if os.environ.get("MOBGS_BENCH_UNSORTED") == "1" else
                                 "rows of both sets stored along a Morton curve (GaussianParams.spatial_sort_(), what "
                                 "densify.TrainableGaussians(keep_sorted) maintains): same Gaussians, permuted")},
    }
    # What the driver's 1 -> N curve must be anchored on: `value` at N = 1 is ONE lean render() per step (BASELINE's
    # metric), `value` at N > 1 counts the 18 renders of a sharded deblur iteration -- unlike quantities.  scale_anchor is
    # the N > 1 workload run on ONE GPU (renders/s): efficiency(N) = value(N) / (N x scale_anchor).
    if world == 1:
        result["scale_anchor"] = deblur["renders_per_s"] if deblur is not None else None
        result["scale_anchor_field"] = "deblur.renders_per_s (this line); compare value of the N > 1 lines with it, not with `value`"
    else:
        result["scale_anchor"] = round(anchor, 2)
        result["scale_anchor_field"] = ("the same 18-unit iteration with every unit on one GPU, timed in this run on every "
                                        "rank (slowest rank): efficiency = value / (n_gpus x scale_anchor)")
        result["scaling_efficiency_vs_anchor"] = round(value / (world * anchor), 4)
        result["scaling_diagnostics"] = scale_diag
    if deblur is not None:
        result["deblur"] = deblur
    if dynamic is not None:
        result["dynamic_config3"] = dynamic
    if flows is not None:
        result["get_flow"] = flows
    if train_it is not None:
        result["train_iteration"] = train_it
    if repeat is not None:
        result["repeat"] = repeat
    if small_scene is not None:
        result["reference_operating_point"] = small_scene
    if rank == 0:
        rb = prof.get("raster_bwd")
        if rb:
            # algorithmic bytes of raster_bwd per launch (DESIGN.md "Kernels"): per intersection 4 (id) + 64 (splat
            # record) gathered + 64 (gradient record) written; per pixel 40 (v_render) + 4 (v_alpha) + 4 (alpha)
            # + 4 (last_id) read
            alg_bytes, _, alg_what = raster_bwd_bytes(I, P)
            achieved = alg_bytes / (rb["avg_ms"] * 1e-3) / 1e9
            roof = {"kernel": "raster_bwd_kernel<10>", "algorithmic_bytes_formula": alg_what, "bound": "hbm",
                    "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "avg_kernel_ms": round(rb["avg_ms"], 4), "algorithmic_bytes": alg_bytes, "calls": rb["calls"]}
            # counter-derived figures are attached only when the committed PMC summary was collected with exactly
            # these kernel sources (scripts/prof_pmc.sh writes profiles/r02_raster_bwd_pmc.json with their hash)
            if args.pmc and world == 1:
                roof["pmc_collected"] = "error" not in collect_pmc(args, I, P)
            pmc = PMC_JSON
            if os.path.exists(pmc):
                try:
                    counters = json.load(open(pmc))
                    if counters.get("source_sha") == source_sha():
                        roof["traffic"] = counters.get("hbm_bytes_per_launch")
                        roof["traffic_provenance"] = (
                            "collected in this run (--pmc)" if roof.get("pmc_collected") else
                            f"committed {os.path.relpath(pmc, ROOT)} (kernel sources sha {str(counters.get('source_sha'))[:12]} = "
                            "this build's; collected by an earlier `bench.py --pmc` run, not by this invocation)")
                        pairs = counters.get("pixel_splat_pairs_per_launch")
                        vi = counters.get("valu_wave_insts_per_launch")
                        if pairs:
                            roof["pixel_splat_pairs_per_s"] = round(pairs / (rb["avg_ms"] * 1e-3), 1)
                        if vi:
                            # VALU lane-operations/s against the fp32 vector peak counted the same way
                            # (157.3 TFLOP/s = 78.6 T lane-ops/s of FMA)
                            lane_ops = vi * 64.0 / (rb["avg_ms"] * 1e-3)
                            roof["valu"] = {"wave_insts": vi, "lane_ops_per_s": round(lane_ops, 1),
                                            "frac_of_fp32_vector_issue_peak": round(lane_ops / (F32_VECTOR_PEAK_TF
                                                                                                 * 1e12 / 2), 4)}
                    else:
                        roof["traffic_note"] = ("profiles/r06_raster_bwd_pmc.json is from other kernel sources: ignored "
                                                "(python bench.py --pmc re-collects it)")
                except Exception:  # noqa: BLE001
                    pass
            if world == 1:
                N = args.ns + args.nd
                # algorithmic bytes of the whole lean step, with the build's culled lists (I) and with
                # gsplat's bounding-box lists (I_box), against the step time
                e2e = {}
                for tag, ii in (("listed_I", I), ("bounding_box_I", I_box)):
                    b = 244.0 * N + 224.0 * ii + 172.0 * P
                    gbs = b / (ms_per_step * 1e-3) / 1e9
                    e2e[tag] = {"intersections": ii, "bytes": b, "GBps": round(gbs, 1),
                                "frac": round(gbs / HBM_PEAK_GBS, 4)}
                roof["end_to_end"] = e2e
                if not args.no_kernel_breakdown:
                    kb = kernel_breakdown(args, N, args.ns, args.nd, I, I_box, P)
                    if kb:
                        roof["streaming"] = kb
            result["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(*raw, scam, args.width, args.height, args.cpu_reps)
            except Exception as exc:  # noqa: BLE001
                result["cpu_baseline"] = {"error": str(exc)}
            if not args.no_cpu_torch:
                try:
                    result["cpu_baseline"]["torch_reference_config1"] = cpu_torch_reference(args.width, args.height)
                except Exception as exc:  # noqa: BLE001
                    result["cpu_baseline"]["torch_reference_config1"] = {"error": str(exc)}
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
