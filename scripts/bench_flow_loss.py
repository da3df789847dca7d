"""The flow-consistency block of a training iteration (train.py:651-671) at the benchmark size: the reference's torch
ops on this GPU against mobgs_amd.loss_utils.flow_warp_loss, forward + backward.  GPU box only:
    python scripts/bench_flow_loss.py [--width 1352 --height 1014 --views 2 --k 9 --steps 10] [--no-torch]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobgs_amd.loss_utils import flow_warp_loss  # noqa: E402
from oracle import render_torch as RT  # noqa: E402  (the reference's statements, here run on the GPU as the baseline)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1352)
    ap.add_argument("--height", type=int, default=1014)
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--k", type=int, default=9)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--no-torch", action="store_true")
    ap.add_argument("--no-image-grads", action="store_true", help="ori / latent detached: no scatter in the backward pass")
    ap.add_argument("--mask-zero", type=float, default=0.0, help="fraction of the masks that is exactly zero")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, K, H, W = a.views, a.k, a.height, a.width
    g = torch.Generator().manual_seed(0)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    pix = torch.stack([xs, ys], dim=-1)

    def coords():
        low = torch.randn(B * K, 2, H // 32, W // 32, generator=g) * 3.0
        s = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=True)
        return (pix + s.permute(0, 2, 3, 1).reshape(B, K, H, W, 2)).contiguous()

    def mask(*shape):
        m = torch.rand(*shape, generator=g)
        if a.mask_zero > 0:
            m = m * (torch.rand(*shape, generator=g) >= a.mask_zero)
        return m

    t = [torch.rand(B, 3, H, W, generator=g), torch.rand(B, K, 3, H, W, generator=g), coords(), coords(),
         mask(B, K, 1, H, W), mask(B, 1, H, W)]
    t = [x.to(dev).requires_grad_(not (a.no_image_grads and i < 2)) for i, x in enumerate(t)]

    def run(fn, **kw):
        for x in t:
            x.grad = None
        loss = fn(*t, **kw)
        loss.backward()
        return loss

    out = {}
    variants = [("hip_combined", flow_warp_loss, dict(combine_taps=True)), ("hip_plain", flow_warp_loss, dict(combine_taps=False))]
    if not a.no_torch:
        variants.append(("torch_reference_ops", RT.flow_warp_loss, {}))
    for name, fn, kw in variants:
        for _ in range(2):
            loss = run(fn, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = run(fn, **kw)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / a.steps * 1e3
        print(f"{name}: {out[name]:.3f} ms per iteration (fwd + bwd, B={B} K={K} {W}x{H}), loss {float(loss):.6f}", flush=True)
    return out


if __name__ == "__main__":
    main()
