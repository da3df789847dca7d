"""Backward pass of the lean render() at 1352x1014 / 300 k splats when the loss leaves part of the image WITHOUT cotangents (a
masked loss): tiles without a valid pixel must cost the backward compositor its prologue only.  ms per backward pass for masks
of 0, 1/3 and 2/3 of the image rows (HIP events around loss.backward(); MOBGS_LIB selects a library variant)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from mobgs_amd.gaussian_renderer import render
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda")
W, H = 1352, 1014
scam, cam, stat, dyn, raw = B.build_scene(dev, 200_000, 100_000, W, H)
bg = torch.zeros(9, device=dev)
g = torch.Generator().manual_seed(100)
v0 = torch.randn(3, H, W, generator=g).to(dev)
params = B.leaves(stat, dyn)
for frac in (0.0, 1 / 3, 2 / 3):
    v = v0.clone()
    v[:, : int(H * frac)] = 0
    ts = []
    for it in range(30):
        for p in params:
            p.grad = None
        out = render(cam, stat, dyn, None, bg)
        loss = (out["render"] * v).sum()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        loss.backward()
        b.record()
        torch.cuda.synchronize()
        if it >= 10:
            ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"rows without cotangents: {frac:.2f} of the image -> backward {ts[len(ts) // 2]:.3f} ms (median of 20)")
