"""Attribute the small PyTorch kernels of one deblur iteration to the aten ops that launch them (torch.profiler on
bench.py's DeblurWorkload).  GPU box only:  python scripts/prof_torch_ops.py [--views 2]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--stacks", action="store_true")
    ap.add_argument("--ns", type=int, default=200_000)
    ap.add_argument("--nd", type=int, default=100_000)
    ap.add_argument("--width", type=int, default=1352)
    ap.add_argument("--height", type=int, default=1014)
    ap.add_argument("--host", action="store_true", help="also list the ops by host (self CPU) time")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from mobgs_amd.distributed import SubframeShard
    torch.autograd.set_multithreading_enabled(False)
    scam, cam, stat, dyn, raw = bench.build_scene(dev, a.ns, a.nd, a.width, a.height)
    wl = bench.DeblurWorkload(dev, stat, dyn, scam, a.width, a.height, SubframeShard(), a.views)
    for _ in range(4):
        wl.step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True,
                 with_stack=a.stacks) as prof:
        wl.step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45,
                                                              max_name_column_width=60))
    if a.host:
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
    print("\n--- PyTorch ops that launch device work (everything except this package's autograd nodes) ---")
    tot = 0.0
    for e in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total):
        if e.self_device_time_total <= 0 or e.key.startswith(("mobgs", "void mobgs", "_", "Prep", "Decode")):
            continue
        if e.key.startswith(("aten::", "Memcpy", "Memset", "autograd", "torch")) or "Backward" in e.key:
            tot += e.self_device_time_total
            print(f"{e.key[:48]:48s} n={e.count:4d} dev={e.self_device_time_total:9.1f}us  {str(e.input_shapes)[:120]}")
    print(f"total {tot:.1f} us")
    if a.stacks:
        print("\n--- the same, by Python call site ---")
        for e in sorted(prof.key_averages(group_by_stack_n=8), key=lambda e: -e.self_device_time_total):
            if e.self_device_time_total <= 0 or not e.key.startswith(("aten::", "Memcpy")):
                continue
            site = [fr for fr in e.stack if "mobgs_amd" in fr or "bench.py" in fr][:3]
            print(f"{e.key[:28]:28s} n={e.count:4d} dev={e.self_device_time_total:8.1f}us  " + " <- ".join(
                fr.split("/")[-1][:60] for fr in site))

if __name__ == "__main__":
    main()
