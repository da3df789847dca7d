#!/usr/bin/env python
"""Which torch ops launch the small non-mobgs kernels of one lean render step (debug aid).
    python scripts/torch_ops.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import mobgs_amd.gaussian_renderer as GR  # noqa: E402

dev = torch.device("cuda:0")
W, H = 1352, 1014
scam, cam, stat, dyn, _ = B.build_scene(dev, 200_000, 100_000, W, H)
bg = torch.zeros(9, device=dev)
g = torch.Generator().manual_seed(100)
v3 = torch.randn(3, H, W, generator=g).to(dev)
v1 = torch.randn(1, H, W, generator=g).to(dev)
params = B.leaves(stat, dyn)


def step():
    for p in params:
        p.grad = None
    out = GR.render(cam, stat, dyn, None, bg)
    torch.autograd.backward([out["render"], out["depth"]], [v3, v1])


for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
evs.sort(key=lambda e: e.time_range.start)
seen = []
for e in evs:
    ks = [k.name[:60] for k in e.kernels]
    if ks and not any(c.kernels for c in e.cpu_children):
        seen.append((e.name, list(e.input_shapes) if e.input_shapes else "", ks))
for s in seen:
    print(s)
for e in evs:
    if e.name in ("aten::copy_", "aten::fill_") and e.kernels:
        print(e.name, e.input_shapes, "\n   ", "\n    ".join(e.stack[:12]))
