#!/usr/bin/env python
"""Heavy-tailed list lengths: the benchmark cloud with a fraction of the splats pulled into a small screen region.
Reports the list-length distribution and the compositing kernel times (profiler regions) beside the uniform scene.
    python scripts/heavy_tail.py [frac_clustered=0.3] [region=0.08]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobgs_amd import profiler, rendering  # noqa: E402
from mobgs_amd.synth import SynthCamera, splat_inputs  # noqa: E402

dev = torch.device("cuda")
cam = SynthCamera()
N = 300000
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
region = float(sys.argv[2]) if len(sys.argv) > 2 else 0.08


def run(clustered):
    s = splat_inputs(N, cam, 0, 9)
    if clustered:
        g = torch.Generator().manual_seed(5)
        k = int(frac * N)
        m = s["means"].clone()
        z = m[:k, 2]
        m[:k, 0] = (torch.rand(k, generator=g) - 0.5) * region * z * cam.width / cam.focal
        m[:k, 1] = (torch.rand(k, generator=g) - 0.5) * region * z * cam.height / cam.focal
        s["means"] = m
    s = {k: v.to(dev) for k, v in s.items()}
    for k in ["means", "quats", "scales", "opacities", "colors", "viewmats"]:
        s[k].requires_grad_(True)
    bg = torch.zeros(1, 9, device=dev)
    g = torch.Generator().manual_seed(100)
    v_img = torch.randn(1, cam.height, cam.width, 10, generator=g).to(dev)

    def step():
        img, a, meta = rendering.rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"],
                                               s["viewmats"], s["Ks"], cam.width, cam.height, packed=False,
                                               backgrounds=bg, render_mode="RGB+ED")
        (img * v_img).sum().backward()
        return meta

    for _ in range(3):
        meta = step()
    torch.cuda.synchronize()
    off = meta["isect_offsets"].reshape(-1)
    lens = torch.cat([off[1:], torch.tensor([meta["flatten_ids"].numel()], device=dev)]) - off
    profiler.enable(True)
    t0 = time.time()
    K = 20
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / K
    prof = profiler.summary()
    profiler.enable(False)
    print(f"{'clustered' if clustered else 'uniform  '} I={meta['flatten_ids'].numel()} mean={lens.float().mean():.0f} "
          f"p99={lens.float().quantile(0.99):.0f} max={int(lens.max())}  step {dt * 1e3:.2f} ms  "
          f"raster_fwd {prof['raster_fwd']['avg_ms']:.3f} ms  raster_bwd {prof['raster_bwd']['avg_ms']:.3f} ms")


run(False)
run(True)
