"""Eager lean render() step (fwd + bwd) at 512x288 / 20 k + 10 k splats: wall clock per step and, with --host, the host
time of the forward and backward halves (the device idles on this scene: the step is host-bound).  A/B with the
environment switches (MOBGS_FUSE_PREP, MOBGS_FUSE_DECODER, MOBGS_FUSED_LISTS, MOBGS_BENCH_UNSORTED, ...)."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from mobgs_amd.gaussian_renderer import render
torch.autograd.set_multithreading_enabled(False)
dev = torch.device('cuda')
W, H = 512, 288
scam, cam, stat, dyn, raw = B.build_scene(dev, 20000, 10000, W, H)
bg = torch.zeros(9, device=dev)
g = torch.Generator().manual_seed(100)
v_render = torch.randn(3, H, W, generator=g).to(dev); v_depth = torch.randn(1, H, W, generator=g).to(dev)
params = B.leaves(stat, dyn)
tf = tb = 0.0
def eager(host=False):
    global tf, tb
    for p in params: p.grad = None
    t0 = time.perf_counter()
    out = render(cam, stat, dyn, None, bg)
    t1 = time.perf_counter()
    torch.autograd.backward([out["render"], out["depth"]], [v_render, v_depth])
    t2 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1
for _ in range(50): eager()
gc.collect(); gc.freeze()
best = 1e9
for rep in range(5):
    tf = tb = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): eager()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 300
    best = min(best, te)
print("eager %.4f ms/step (best of 5 x 300); host: forward %.1f us, backward %.1f us" % (best * 1e3, tf / 300 * 1e6, tb / 300 * 1e6))
