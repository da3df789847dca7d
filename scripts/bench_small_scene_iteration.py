"""The K = 9 deblur iteration of two views (train.py:430-541) at the reference's own operating point (512x288, 20 k + 10 k
splats) and at the headline size: one render() per sub-frame against one render_many() batch per view."""
import gc, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from mobgs_amd.distributed import SubframeShard
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda")
for (W, H, ns, nd, steps) in ((512, 288, 20_000, 10_000, 40), (1352, 1014, 200_000, 100_000, 10)):
    scam, cam, stat, dyn, raw = B.build_scene(dev, ns, nd, W, H)
    res = {}
    for batched in (False, True):
        wl = B.DeblurWorkload(dev, stat, dyn, scam, W, H, SubframeShard(1, 0), 2, batched=batched)
        wl.step()
        gc.collect(); gc.freeze()   # (a generation-2 collection costs ~66 ms on this host: see bench.py timed())
        for _ in range(5):
            wl.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            wl.step()
        torch.cuda.synchronize()
        res[batched] = (time.perf_counter() - t0) / steps * 1e3
    print(f"{W}x{H} {ns}+{nd}: K=9 two-view iteration  separate renders {res[False]:.3f} ms   batched sub-frames {res[True]:.3f} ms")

# ... and the WHOLE training iteration (train.py:430-807: both blurry views, 18 get_flow calls, losses, densification
# statistics, Adam) at the reference's operating point
sys.path.insert(0, os.path.join(ROOT, "examples"))
import train_deblur_synth as TD
tr = TD.DeblurTrainer("cuda:0", 20_000, 10_000, 512, 288, 2, iters=10000)
tr.iteration()
gc.collect(); gc.freeze()
for _ in range(5):
    tr.iteration()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30):
    tr.iteration()
torch.cuda.synchronize()
print(f"512x288 20000+10000: whole training iteration {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms")
