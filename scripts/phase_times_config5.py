"""Where does the K = 9 deblur iteration at 800 k splats spend its time (HIP events around the phases, no profiler)?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from mobgs_amd.distributed import SubframeShard
from mobgs_amd import profiler, deblur
import mobgs_amd.gaussian_renderer as GR
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda:0")
ns, nd = (533_000, 267_000) if "--small" not in sys.argv else (200_000, 100_000)
scam, cam, stat, dyn, _ = B.build_scene(dev, ns, nd, 1352, 1014, seed=1)
batched = "--separate" not in sys.argv
wl = B.DeblurWorkload(dev, stat, dyn, scam, 1352, 1014, SubframeShard(world_size=1, rank=0), batched=batched)
orig_many, orig_render = GR.render_many, GR.render
def many(*a, **k):
    with profiler.region("render_many fwd"):
        return orig_many(*a, **k)
def rend(*a, **k):
    with profiler.region("render fwd"):
        return orig_render(*a, **k)
GR.render_many = many
deblur.render = rend
import mobgs_amd.gaussian_renderer
orig_bw = torch.autograd.backward
def bw(*a, **k):
    with profiler.region("backward"):
        return orig_bw(*a, **k)
torch.autograd.backward = bw
if "--lean-first" in sys.argv:
    from mobgs_amd.ops import LeafGradSink
    bg = torch.zeros(9, device=dev)
    g = torch.Generator().manual_seed(100)
    v3, v1 = torch.randn(3, 1014, 1352, generator=g).to(dev), torch.randn(1, 1014, 1352, generator=g).to(dev)
    for _ in range(120):
        out = orig_render(cam, stat, dyn, None, bg)
        with LeafGradSink(stat, dyn):
            orig_bw([out["render"], out["depth"]], [v3, v1])
    torch.cuda.synchronize()
for _ in range(6):
    wl.step()
torch.cuda.synchronize()
profiler.enable(True)
t0 = time.perf_counter()
NIT = 12 if "--long" in sys.argv else 5
for _ in range(NIT):
    with profiler.region("whole step"):
        wl.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / NIT * 1e3
s = profiler.summary()
from mobgs_amd import rendering
print("batched" if batched else "separate", ns + nd, "wall %.2f ms/iteration" % dt, "list rebuilds so far", rendering.list_rebuilds[0])
per = [a.elapsed_time(b) for a, b in profiler._events["whole step"]]
print("  per-iteration ms:", [round(x, 1) for x in per])
print("  render_many fwd ms:", [round(a.elapsed_time(b), 1) for a, b in profiler._events["render_many fwd"]])
for k, v in s.items():
    print("  %-18s calls/it %5.1f  ms/it %7.2f" % (k, v["calls"] / NIT, v["total_ms"] / NIT))
