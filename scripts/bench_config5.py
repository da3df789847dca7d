"""BASELINE config #5 timed on one GPU: 800 000 Gaussians (533 k static + 267 k dynamic), attributes stored as fp16 and
read by the kernels as halves (fp32 masters for the optimiser: DESIGN 7a), 1352x1014 -- the lean render step and the
K = 9 deblur iteration of two views (BLCE cameras), forward + backward.  GPU box only:
    python scripts/bench_config5.py [--steps 10] [--fp32]"""
import argparse
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from mobgs_amd.camera import PinholeCamera  # noqa: E402
from mobgs_amd.distributed import SubframeShard  # noqa: E402
from mobgs_amd.gaussian_model import GaussianParams  # noqa: E402
from mobgs_amd.gaussian_renderer import render  # noqa: E402
from mobgs_amd.helper_model import Sandwich  # noqa: E402
from mobgs_amd.ops import LeafGradSink  # noqa: E402
from mobgs_amd.synth import SynthCamera, dynamic_extras, gaussian_cloud  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--fp32", action="store_true", help="fp32 attribute storage (the twin scene) for comparison")
    ap.add_argument("--separate", action="store_true", help="one render() per latent sub-frame instead of render_many")
    a = ap.parse_args()
    torch.autograd.set_multithreading_enabled(False)
    dev = torch.device("cuda:0")
    W, H, ns, nd = 1352, 1014, 533_000, 267_000
    scam = SynthCamera()
    sp, dp = gaussian_cloud(ns, scam, 1), gaussian_cloud(nd, scam, 2)
    dx = dynamic_extras(dp["xyz"], 1)
    torch.manual_seed(1)
    dec = Sandwich(9, 3).to(dev)
    dt = torch.float32 if a.fp32 else torch.float16
    stat = GaussianParams(sp, None, dec, dev, requires_grad=True, attr_dtype=dt)
    dyn = GaussianParams(dp, dx, dec, dev, requires_grad=True, attr_dtype=dt)
    if not a.fp32:
        stat.enable_fp32_masters(False)
        dyn.enable_fp32_masters(True)
    cam = PinholeCamera(W, H, scam.K, torch.eye(4), time=scam.time, max_time=scam.max_time, device=dev)
    bg = torch.zeros(9, device=dev)
    g = torch.Generator().manual_seed(100)
    v3, v1 = torch.randn(3, H, W, generator=g).to(dev), torch.randn(1, H, W, generator=g).to(dev)
    params = B.leaves(stat, dyn)

    def lean():
        for p in params:
            p.grad = None
        out = render(cam, stat, dyn, None, bg)
        with LeafGradSink(stat, dyn):
            torch.autograd.backward([out["render"], out["depth"]], [v3, v1])

    def timed(fn, steps, warm):
        fn()
        gc.collect()
        gc.freeze()   # (a generation-2 collection costs ~66 ms on this host: see bench.py timed())
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    tag = "fp32 attributes" if a.fp32 else "fp16 attributes + fp32 masters"
    ms = timed(lean, 10 * a.steps, 20)
    print(f"config #5 ({ns + nd} Gaussians, {W}x{H}, {tag}): lean render fwd+bwd {ms:.3f} ms = {1e3 / ms:.1f} renders/s",
          flush=True)
    wl = B.DeblurWorkload(dev, stat, dyn, scam, W, H, SubframeShard(world_size=1, rank=0), batched=not a.separate)
    for _ in range(6):
        wl.step()
    st0 = torch.cuda.memory_stats()
    ms = timed(wl.step, a.steps, 0)
    st1 = torch.cuda.memory_stats()
    from mobgs_amd import profiler
    profiler.enable(True)
    for _ in range(3):
        wl.step()
    summ = profiler.summary()
    profiler.enable(False)
    print("HIP-event regions over 3 iterations:", {k: (v["calls"], round(v["total_ms"], 2)) for k, v in summ.items()},
          flush=True)
    print("allocator during the timed iterations: device mallocs", st1["num_device_alloc"] - st0["num_device_alloc"],
          "device frees", st1["num_device_free"] - st0["num_device_free"], "retries",
          st1["num_alloc_retries"] - st0["num_alloc_retries"], "reserved GiB", st1["reserved_bytes.all.current"] / 2**30,
          flush=True)
    print(f"config #5: K = 9 deblur iteration of two views {ms:.2f} ms = {18e3 / ms:.0f} renders/s, "
          f"{2e3 / ms:.1f} blurry views/s; memory in use {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)


if __name__ == "__main__":
    main()
