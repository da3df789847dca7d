"""VERDICT r5 item 4 ("find the 40 ms"): ONE whole training iteration at the headline size, timed per iteration with the
host clock (a synchronisation per iteration: this is a probe, not the benchmark), next to everything that can stall a step:
list rebuilds / key-segment overflows (rendering), the caching allocator's device allocations and retries, Python
collections, enumeration-order refreshes.  usage: python scripts/r06/train_iter_probe.py [iters] [warmup] [unchanged]"""
import gc
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch

import train_deblur_synth as TD
from mobgs_amd import rendering as R

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
unchanged = len(sys.argv) > 3 and sys.argv[3] == "unchanged"
torch.autograd.set_multithreading_enabled(False)
tr = TD.DeblurTrainer("cuda:0", 200_000, 100_000, 1352, 1014, 2, iters=10000)
step = tr.iteration_unchanged if unchanged else tr.iteration


def snap():
    ms = torch.cuda.memory_stats()
    return dict(rebuilds=R.list_rebuilds[0], seg_over=R.seg_overflows[0], dev_alloc=ms.get("num_device_alloc", 0),
                dev_free=ms.get("num_device_free", 0), retries=ms.get("num_alloc_retries", 0),
                reserved_mb=ms.get("reserved_bytes.all.current", 0) >> 20, gc2=gc.get_stats()[2]["collections"])


rows = []
for i in range(warm + iters):
    if i == 1:
        gc.collect()
        gc.freeze()
    torch.cuda.synchronize()
    a = snap()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    b = snap()
    rows.append((i, dt, {k: b[k] - a[k] for k in a if k != "reserved_mb"}, b["reserved_mb"]))
for i, dt, d, res in rows:
    flag = " ".join(f"{k}+{v}" for k, v in d.items() if v)
    print(f"iter {i:3d}{' (warm-up)' if i < warm else ''}: {dt:8.2f} ms  reserved {res} MB  {flag}")
ts = sorted(dt for i, dt, _, _ in rows if i >= warm)
print(f"timed {len(ts)}: mean {sum(ts) / len(ts):.2f} median {ts[len(ts) // 2]:.2f} max {ts[-1]:.2f} min {ts[0]:.2f} ms")
