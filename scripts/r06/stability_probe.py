"""Stability over a longer run: 3000 lean steps and 300 graphed / 300 eager training iterations at 512x288 -- time per step in windows,
allocator state at the start and at the end (a leak or an arena that keeps growing would show)."""
import gc, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import bench as B
import train_deblur_synth as TD
from mobgs_amd.gaussian_renderer import render
from mobgs_amd.graphed import GraphedCallable
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda")


def mem():
    s = torch.cuda.memory_stats()
    return f"allocated {torch.cuda.memory_allocated() >> 20} MB, reserved {torch.cuda.memory_reserved() >> 20} MB, device mallocs {s.get('num_device_alloc', 0)}"


scam, cam, stat, dyn, raw = B.build_scene(dev, 200_000, 100_000, 1352, 1014)
bg = torch.zeros(9, device=dev)
g = torch.Generator().manual_seed(100)
v = torch.randn(3, 1014, 1352, generator=g).to(dev)
params = B.leaves(stat, dyn)


def lean():
    for p in params:
        p.grad = None
    out = render(cam, stat, dyn, None, bg)
    ((out["render"] * v).sum() + out["depth"].sum()).backward()


for _ in range(50):
    lean()
gc.collect(); gc.freeze(); torch.cuda.synchronize()
print("lean step, 1352x1014 / 300 k:", mem())
for w in range(6):
    t0 = time.perf_counter()
    for _ in range(500):
        lean()
    torch.cuda.synchronize()
    print(f"  steps {500 * w:4d}..{500 * w + 499}: {(time.perf_counter() - t0) / 500 * 1e3:.4f} ms per step")
print("  end:", mem())
for graph in (False, True):
    tr = TD.DeblurTrainer("cuda:0", 20_000, 10_000, 512, 288, 2, iters=10000, lambda_flow=0.0)
    tr.iteration()
    fb = GraphedCallable(tr.forward_backward, warmup=0) if graph else None
    torch.cuda.synchronize()
    print(f"training loop 512x288 / 30 k ({'graphed' if graph else 'eager'}):", mem())
    losses = []
    recaptures = [0]
    pending = []
    for w in range(3):
        t0 = time.perf_counter()
        for _ in range(100):
            if fb is not None:
                losses.append(fb()); tr.optimizer_step()
                ev = torch.cuda.Event(); ev.record(); pending.append(ev)
                if len(pending) > 2:
                    pending.pop(0).synchronize()   # the host stays at most two iterations ahead
                if not fb.check():     # (sees the completed replays) an arena outgrown: record again
                    torch.cuda.synchronize(); fb.recapture(); recaptures[0] += 1
            else:
                losses.append(tr.iteration())
        torch.cuda.synchronize()
        ok = fb.check() if fb is not None else True
        print(f"  iterations {100 * w:3d}..{100 * w + 99}: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms per iteration, loss {float(losses[-1]):.5f}, arenas fitted {ok}, recaptures so far {recaptures[0]}")
    print("  end:", mem())
    del tr, fb
