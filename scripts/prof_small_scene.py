"""Host-side profile of render() fwd+bwd at the reference's own operating point (512x288, ~30 k splats,
scene/dataset_readers.py:1448-1460), where the step is bound by the host, not the device.  GPU box only:
    python scripts/prof_small_scene.py [--ns 20000 --nd 10000 --width 512 --height 288] [--cprofile]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mobgs_amd.gaussian_renderer as GR  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ns", type=int, default=20000)
    ap.add_argument("--nd", type=int, default=10000)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=288)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--cprofile", action="store_true")
    ap.add_argument("--heavy", type=int, default=-1, help="rendering.tuning.heavy_tile_len (-1: library default)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--dense", action="store_true", help="experiment: force the dense-region binning variant")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.autograd.set_multithreading_enabled(False)  # backward on the calling thread
    from mobgs_amd import rendering
    rendering.tuning.heavy_tile_len = a.heavy
    if a.dense:
        class _Hint(dict):
            def get(self, k, d=None):
                return 4096
        rendering._len_hint = _Hint()
    scam, cam, stat, dyn, _ = bench.build_scene(dev, a.ns, a.nd, a.width, a.height)
    bg = torch.zeros(9, device=dev)
    g = torch.Generator().manual_seed(100)
    v3 = torch.randn(3, a.height, a.width, generator=g).to(dev)
    v1 = torch.randn(1, a.height, a.width, generator=g).to(dev)
    params = bench.leaves(stat, dyn)

    def step():
        for p in params:
            p.grad = None
        out = GR.render(cam, stat, dyn, None, bg)
        torch.autograd.backward([out["render"], out["depth"]], [v3, v1])

    def fwd_only():
        with torch.no_grad():
            GR.render(cam, stat, dyn, None, bg)["render"]

    for name, fn in (("fwd+bwd", step), ("fwd only (no_grad)", fwd_only)):
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        t_host = time.perf_counter() - t0  # all launches issued
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"{name}: {t_all / a.steps * 1e3:.3f} ms/step wall, host issue time {t_host / a.steps * 1e3:.3f} ms/step")
    from mobgs_amd import rendering as R
    print("intersections", R.last_stats.get("n_isects"), "longest list", R.last_stats.get("max_tile_len"))
    if a.no_profile:
        return
    # device time of one step
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    ev = prof.key_averages()
    dev_us = sum(e.self_device_time_total for e in ev) / 10
    print(f"device time per step: {dev_us:.1f} us")
    print(ev.table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=50))
    if a.cprofile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(200):
            step()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(35)


if __name__ == "__main__":
    main()
