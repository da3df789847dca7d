"""Kernel-profile driver: the K = 9 two-view deblur iteration at 512x288 / 20 k + 10 k splats (batched sub-frames)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from mobgs_amd.distributed import SubframeShard
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda")
W, H = 512, 288
scam, cam, stat, dyn, raw = B.build_scene(dev, 20_000, 10_000, W, H)
wl = B.DeblurWorkload(dev, stat, dyn, scam, W, H, SubframeShard(1, 0), 2)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    wl.step()
torch.cuda.synchronize()
