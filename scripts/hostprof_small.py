"""cProfile of the host side of a lean render() step on a small scene (the step is host-bound there)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import mobgs_amd.gaussian_renderer as GR

dev = torch.device("cuda:0")
W, H = 1352, 1014
scam, cam, stat, dyn, _ = B.build_scene(dev, 20000, 10000, W, H)
bg = torch.zeros(9, device=dev)
g = torch.Generator().manual_seed(100)
v3 = torch.randn(3, H, W, generator=g).to(dev)
v1 = torch.randn(1, H, W, generator=g).to(dev)
params = B.leaves(stat, dyn)


def step():
    for p in params:
        p.grad = None
    out = GR.render(cam, stat, dyn, None, bg)
    torch.autograd.backward([out["render"], out["depth"]], [v3, v1])


for _ in range(30):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(300):
    step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 300 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
