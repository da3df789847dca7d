"""Host-side stall hunt: time (host clock) the pieces of render_many over many iterations at 800 k splats, report outliers."""
import os, sys, time, gc, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from mobgs_amd.distributed import SubframeShard
from mobgs_amd import rendering as R
import mobgs_amd.gaussian_renderer as GR
import mobgs_amd.ops as OPS
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda:0")
scam, cam, stat, dyn, _ = B.build_scene(dev, 533_000, 267_000, 1352, 1014, seed=1)
wl = B.DeblurWorkload(dev, stat, dyn, scam, 1352, 1014, SubframeShard(world_size=1, rank=0), batched=True)
log = []
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            log.append((tag, (time.perf_counter() - t0) * 1e3))
    setattr(obj, name, g)
wrap(GR, "_prep", "prep")
wrap(R.SharedProjection, "__init__", "SharedProjection")
wrap(R.SharedProjection, "composite", "composite")
wrap(GR, "decode", "decode")
wrap(R.TileLists, "resolve", "resolve")
wrap(R._PendingCounts, "_wait", "wait_counts")
wrap(GR, "render_many", "render_many")
gc_events = []
def on_gc(phase, info):
    if phase == "start":
        on_gc.t0 = time.perf_counter()
    else:
        gc_events.append((info["generation"], (time.perf_counter() - on_gc.t0) * 1e3, info["collected"]))
gc.callbacks.append(on_gc)
for it in range(30):
    n0 = len(log)
    t0 = time.perf_counter()
    wl.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    big = [(t, round(ms, 1)) for t, ms in log[n0:] if ms > 15]
    if dt > 35 or big:
        print(f"iteration {it}: {dt:.1f} ms; host pieces > 15 ms: {big}; gc so far: {[(g, round(ms, 1), c) for g, ms, c in gc_events if ms > 5]}", flush=True)
print("gc events (gen, ms, collected) > 5 ms:", [(g, round(ms, 1), c) for g, ms, c in gc_events if ms > 5])
print("tracked objects:", len(gc.get_objects()))
