"""The 9 get_flow() calls of one view (train.py:570-579), forward + backward, on the benchmark scene.  GPU box only:
    python scripts/prof_flow.py [--steps 5] [--zero-weight] [--separate]      (under scripts/prof.sh for kernel stats)
--zero-weight: cotangents exactly zero (lambda_flow_loss = 0, the seesaw configuration); --separate: nine get_flow
calls as train.py issues them instead of one get_flow_many."""
import argparse
import contextlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mobgs_amd.gaussian_renderer as GR  # noqa: E402
from mobgs_amd.ops import LeafGradSink  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--zero-weight", action="store_true")
    ap.add_argument("--separate", action="store_true")
    ap.add_argument("--no-sink", action="store_true", help="plain loss.backward() without ops.LeafGradSink")
    ap.add_argument("--torch-ops", action="store_true", help="list the PyTorch ops of one step that launch device work")
    ap.add_argument("--ns", type=int, default=200_000)
    ap.add_argument("--nd", type=int, default=100_000)
    ap.add_argument("--width", type=int, default=1352)
    ap.add_argument("--height", type=int, default=1014)
    a = ap.parse_args()
    torch.autograd.set_multithreading_enabled(False)
    dev = torch.device("cuda:0")
    W, H = a.width, a.height
    scam, cam, stat, dyn, _ = bench.build_scene(dev, a.ns, a.nd, W, H)
    bg = torch.zeros(9, device=dev)
    g = torch.Generator().manual_seed(100)
    v3, v1 = torch.randn(3, H, W, generator=g).to(dev), torch.randn(1, H, W, generator=g).to(dev)
    v2 = torch.randn(1, H, W, 2, generator=g).to(dev)
    if a.zero_weight:
        v3, v1, v2 = torch.zeros_like(v3), torch.zeros_like(v1), torch.zeros_like(v2)
    deltas = [float(d) for d in torch.linspace(-1.0, 1.0, 9)]  # train.py:571-573: (k - half) / half
    params = bench.leaves(stat, dyn)

    def step():
        for p in params:
            p.grad = None
        GR.invalidate_flow_cache()   # (no optimiser step here: without this the implicit mid-state cache would hit across steps)
        if a.separate:
            outs = [GR.get_flow(cam, stat, dyn, None, bg, delta_exposure=d) for d in deltas]
        else:
            outs = GR.get_flow_many(cam, stat, dyn, None, bg, deltas)
        with (contextlib.nullcontext() if a.no_sink else LeafGradSink(stat, dyn)):
            torch.autograd.backward([t for o in outs for t in o], [v2, v2, v3, v1] * 9)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if a.torch_ops:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            step()
            torch.cuda.synchronize()
        tot = 0.0
        for e in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total):
            if e.self_device_time_total > 0 and (e.key.startswith(("aten::", "Memcpy", "Memset"))):
                tot += e.self_device_time_total
                print(f"{e.key[:40]:40s} n={e.count:4d} dev={e.self_device_time_total:9.1f}us  {str(e.input_shapes)[:110]}")
        print(f"total {tot:.1f} us")
    print(f"9 x get_flow fwd+bwd: {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms per view "
          f"({'zero' if a.zero_weight else 'random'} cotangents, {'separate calls' if a.separate else 'get_flow_many'})")


if __name__ == "__main__":
    main()
