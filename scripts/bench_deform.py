#!/usr/bin/env python
"""BASELINE config #3 leg: `deform_network` (HexPlane + MLP heads) at the seesaw plane resolution
(arguments/stereo/seesaw.py: [64,64,64,12] x multires [1,2,4], 32 channels) on the dynamic points of the benchmark
scene.  Per-kernel HIP-event times, forward and forward+backward, with the fp32-MFMA and HBM figures DESIGN.md quotes.

    python scripts/bench_deform.py [--n 100000] [--steps 20]
Prints one JSON object; not part of the bench.py contract.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobgs_amd.deformation import SeesawArgs, deform_network  # noqa: E402
from mobgs_amd.synth import SynthCamera, gaussian_cloud  # noqa: E402

MLP_MACS_PER_POINT = 96 * 128 + 3 * (128 * 128) + 128 * 14  # 63 232
F32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at the fp32 vector rate


def event_ms(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    evs = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[100_000, 300_000])
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {}
    for n in args.n:
        torch.manual_seed(0)
        net = deform_network(SeesawArgs()).to(dev)
        cloud = gaussian_cloud(n, SynthCamera(), 1)
        pts = cloud["xyz"].to(dev).requires_grad_(True)
        scales = cloud["scaling"].to(dev).requires_grad_(True)
        rots = cloud["rotation"].to(dev).requires_grad_(True)
        lo, hi = cloud["xyz"].min(0).values, cloud["xyz"].max(0).values
        net.deformation_net.set_aabb(hi.tolist(), lo.tolist())
        with torch.no_grad():
            for pl in net.deformation_net.grid.planes():
                pl.uniform_(0.5, 1.0)
        times = torch.full((n, 1), 11.0 / 23.0, device=dev)
        g = torch.Generator().manual_seed(3)
        cot = [torch.randn(n, k, generator=g).to(dev) for k in (3, 3, 4)]
        params = list(net.parameters())

        def fwd():
            with torch.no_grad():
                return net(pts, scales, rots, times)

        def fwd_bwd():
            for p in params:
                p.grad = None
            pts.grad = scales.grad = rots.grad = None
            o = net(pts, scales, rots, times)
            torch.autograd.backward(o, cot)

        r = {"fwd_ms": event_ms(fwd, args.steps), "fwd_bwd_ms": event_ms(fwd_bwd, args.steps)}
        # the four kernels by themselves
        from mobgs_amd import deformation as D
        r.update(D.kernel_times(net, pts.detach(), scales.detach(), rots.detach(), times, cot, args.steps, event_ms))
        flops_fwd = 2.0 * MLP_MACS_PER_POINT * n
        if "mlp_fwd_ms" in r:
            r["mlp_fwd_tflops"] = round(flops_fwd / (r["mlp_fwd_ms"] * 1e-3) / 1e12, 2)
            r["mlp_fwd_frac_of_f32_mfma_peak"] = round(r["mlp_fwd_tflops"] / F32_MFMA_PEAK_TF, 3)
        if "mlp_bwd_ms" in r:
            # backward = recomputed forward of the three hidden layers + data gradients + weight gradients
            flops_bwd = 2.0 * n * (MLP_MACS_PER_POINT - 128 * 14 + 2 * MLP_MACS_PER_POINT)
            r["mlp_bwd_tflops"] = round(flops_bwd / (r["mlp_bwd_ms"] * 1e-3) / 1e12, 2)
            r["mlp_bwd_frac_of_f32_mfma_peak"] = round(r["mlp_bwd_tflops"] / F32_MFMA_PEAK_TF, 3)
        out[str(n)] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
