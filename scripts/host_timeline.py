"""Where does the HOST spend a bench step? (perf_counter around the phases; no device syncs added)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from mobgs_amd.gaussian_renderer import render
import mobgs_amd.rendering as R
dev = torch.device("cuda:0")
W, H = 1352, 1014
scam, cam, stat, dyn, _ = B.build_scene(dev, 200_000, 100_000, W, H)
bg = torch.zeros(9, device=dev)
g = torch.Generator().manual_seed(100)
v3 = torch.randn(3, H, W, generator=g).to(dev); v1 = torch.randn(1, H, W, generator=g).to(dev)
params = B.leaves(stat, dyn)
marks = []
orig = R._ProjectAndBin.forward
def timed_fwd(ctx, *a):
    t0 = time.perf_counter(); r = orig(ctx, *a); marks.append(("project_and_bin", t0, time.perf_counter())); return r
R._ProjectAndBin.forward = staticmethod(timed_fwd)
def step():
    t0 = time.perf_counter()
    for p in params: p.grad = None
    out = render(cam, stat, dyn, None, bg)
    t1 = time.perf_counter()
    torch.autograd.backward([out["render"], out["depth"]], [v3, v1])
    t2 = time.perf_counter()
    return t0, t1, t2
for _ in range(10): step()
torch.cuda.synchronize()
rows = []
for _ in range(8):
    marks.clear()
    t0, t1, t2 = step()
    pb = marks[0]
    rows.append((t0, pb[1] - t0, pb[2] - pb[1], t1 - pb[2], t2 - t1))
torch.cuda.synchronize()
prev = None
for t0, a, b, c, d in rows:
    gap = 0 if prev is None else (t0 - prev) * 1e6
    print(f"step start +{gap:7.1f} us | before project_and_bin {a*1e6:6.1f} | project_and_bin (incl. sync) {b*1e6:7.1f} | rest of fwd {c*1e6:6.1f} | backward launch {d*1e6:6.1f}")
    prev = t0
