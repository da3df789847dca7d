"""The WHOLE training iteration (train.py:430-807) at the reference's own operating point -- 512x288, 30 k splats -- with its
forward + loss + backward recorded ONCE as a HIP graph (mobgs_amd.graphed.GraphedCallable) and the one-launch Adam step
outside: eager vs graphed time per iteration, and the gradient buffer of a graphed iteration against the eager one on the
same parameters (bit for bit).   python scripts/r06/graph_small_iteration.py [W H ns nd]"""
import gc, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import train_deblur_synth as TD
from mobgs_amd.graphed import GraphedCallable

W, H, ns, nd = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (512, 288, 20_000, 10_000)
lam = float(os.environ.get("LAMBDA_FLOW", "1e-2"))


def timed(step, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


tr = TD.DeblurTrainer("cuda:0", ns, nd, W, H, 2, iters=10000, lambda_flow=lam)
tr.iteration()
gc.collect(); gc.freeze()
for _ in range(5):
    tr.iteration()
eager_ms = timed(tr.iteration)
eager_fb_ms = timed(tr.forward_backward)

# bit-identity: one eager forward_backward and one graphed one on the SAME parameters (and eager against eager: is the
# iteration itself run-to-run deterministic?)
tr.forward_backward()
torch.cuda.synchronize()
ref = tr.bucket.flat.clone()
tr.forward_backward()
torch.cuda.synchronize()
d = (ref - tr.bucket.flat).abs()
print(f"eager vs eager: equal {bool(torch.equal(ref, tr.bucket.flat))}, max |diff| {float(d.max())}, differing words {int((d > 0).sum())} of {d.numel()}")


def where(diff):
    idx = torch.nonzero(diff > 0).flatten()
    if idx.numel() == 0:
        return "-"
    off, names = 0, []
    lo, hi = int(idx.min()), int(idx.max())
    for p_ in tr.bucket.params:
        n = p_.numel()
        if off + n > lo and off <= hi and bool((diff[off:off + n] > 0).any()):
            names.append(f"{tuple(p_.shape)}@{off}")
        off += n
    return f"words {lo}..{hi} (param floats {tr.bucket.param_floats}): " + ", ".join(names[:8])


fb = GraphedCallable(tr.forward_backward, warmup=0)
loss = fb()            # warm-up (eager) + capture
fb()                   # a replay
torch.cuda.synchronize()
ok = fb.check()
same = bool(torch.equal(ref, tr.bucket.flat)) if ref is not None else None
maxdiff = float((ref - tr.bucket.flat).abs().max()) if ref is not None else None
print(f"capture ok, arenas fitted: {ok}; gradient buffer after a replay == eager: {same} (max |diff| {maxdiff}); where: {where((ref - tr.bucket.flat).abs())}")


def graphed_iteration():
    fb()
    tr.optimizer_step()


for _ in range(5):
    graphed_iteration()
graph_ms = timed(graphed_iteration)
graph_fb_ms = timed(fb)
print(f"{W}x{H} {ns}+{nd} lambda_flow {lam}: whole iteration eager {eager_ms:.3f} ms (forward+backward part {eager_fb_ms:.3f}) -> "
      f"graphed {graph_ms:.3f} ms (forward+backward replay {graph_fb_ms:.3f}); arenas fitted: {fb.check()}")
# the parameters moved through 40 Adam steps meanwhile: the recorded graph on the CURRENT parameters against eager
fb()
torch.cuda.synchronize()
g1 = tr.bucket.flat.clone()
tr.forward_backward()
torch.cuda.synchronize()
print("after 40 optimiser steps: graphed == eager:", bool(torch.equal(g1, tr.bucket.flat)),
      "max |diff|", float((g1 - tr.bucket.flat).abs().max()), "of max", float(g1.abs().max()))
