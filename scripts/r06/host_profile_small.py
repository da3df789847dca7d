"""cProfile of the eager lean step at 512x288 / 30 k splats (host-bound there): where the host's ~350 us per step go."""
import cProfile, gc, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
def eager():
    for p in params: p.grad = None
    out = render(cam, stat, dyn, None, bg)
    torch.autograd.backward([out["render"], out["depth"]], [v_render, v_depth])
for _ in range(50): eager()
gc.collect(); gc.freeze()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(500): eager()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
