"""The graphed training loop at 512x288 / 30 k over several hundred iterations: does a replay survive the scene moving under it?
(iterations in windows of 20, check() after each; env switches select paths for bisection)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import train_deblur_synth as TD
import mobgs_amd.rendering as R
from mobgs_amd.graphed import GraphedCallable
torch.autograd.set_multithreading_enabled(False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
margin = float(os.environ.get("MARGIN", "1.5"))
tr = TD.DeblurTrainer("cuda:0", 20_000, 10_000, 512, 288, 2, iters=10000, lambda_flow=0.0)
tr.iteration()
fb = GraphedCallable(tr.forward_backward, warmup=0, margin=margin)
for w in range(n // 20):
    for _ in range(20):
        loss = fb(); tr.optimizer_step()
    torch.cuda.synchronize()
    ok = fb.check()
    rows = [(int(r[0][0]), int(r[0][1]), int(r[0][2]), r[2], r[3], r[5]) for r in fb.static.rows]
    print(f"iteration {20 * w + 19}: loss {float(loss):.5f} fitted {ok}; (I_box, I, longest | cap_box, cap_listed, seg_stride) per workload: {rows}", flush=True)
    if not ok:
        print("  -> recapture", flush=True)
        fb.recapture()
print("done", flush=True)
