import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench as B
import mobgs_amd.gaussian_renderer as GR
dev = torch.device("cuda:0")
W, H = 1352, 1014
for ns, nd in ((20000, 10000), (200000, 100000)):
    scam, cam, stat, dyn, _ = B.build_scene(dev, ns, nd, W, H)
    bg = torch.zeros(9, device=dev)
    g = torch.Generator().manual_seed(100)
    v3 = torch.randn(3, H, W, generator=g).to(dev); v1 = torch.randn(1, H, W, generator=g).to(dev)
    params = B.leaves(stat, dyn)
    def step():
        for p in params: p.grad = None
        out = GR.render(cam, stat, dyn, None, bg)
        torch.autograd.backward([out["render"], out["depth"]], [v3, v1])
    for mt in (True, False, True, False):
        with torch.autograd.set_multithreading_enabled(mt):
            for _ in range(30): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(300): step()
            torch.cuda.synchronize()
            print(ns + nd, "multithreading", mt, "ms/step %.4f" % ((time.perf_counter() - t0) / 300 * 1e3))
