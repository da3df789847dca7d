import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench as B
from mobgs_amd.graphed import GraphedRenderStep
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
    return out
for _ in range(20): eager()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): out_e = eager()
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 300
ref = {k: out_e[k].detach().clone() for k in ("render", "depth")}; gref = [p.grad.clone() for p in params]
del out_e
import gc; gc.collect()
step = GraphedRenderStep(stat, dyn, W, H, scam.K, bg)
w2c = torch.eye(4)
step.capture(w2c, scam.time)
out = step(w2c, scam.time, v_render, v_depth)
torch.cuda.synchronize()
print("fits:", step.check())
print("render equal:", torch.equal(out["render"], ref["render"]), "depth equal:", torch.equal(out["depth"], ref["depth"]))
print("grads equal:", [torch.equal(p.grad, gr) for p, gr in zip(params, gref)])
for _ in range(20): step(w2c, scam.time)
torch.cuda.synchronize(); t0 = time.perf_counter()
states = [step.camera_state(B.view_pose(i), (5.0 + i) / 23.0) for i in range(4)]
for i in range(300): step(state=states[i % 4])
torch.cuda.synchronize(); tg1 = (time.perf_counter() - t0) / 300
print("graphed with a (cached) camera change per step: %.4f ms/step" % (tg1 * 1e3))
t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 300
print("eager %.4f ms/step, graphed %.4f ms/step" % (te * 1e3, tg * 1e3))
