import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_flow_loss as T
from oracle import render_torch as RT
from mobgs_amd.loss_utils import flow_warp_loss
dev = torch.device("cuda:0")
for (B, K, H, W, seed) in [(1, 1, 5, 7, 0), (2, 3, 37, 70, 1), (1, 9, 67, 129, 2), (2, 2, 130, 64, 3), (1, 3, 1014, 1352, 9)]:
    case = T._case(B, K, H, W, seed, flow=3.0 if W > 1000 else 2.5)
    ref, gref = T._run(RT.flow_warp_loss, case, "cpu")
    got, ggot = T._run(flow_warp_loss, case, dev)
    line = [f"{B}x{K}x{H}x{W}: loss rel err {abs(got - ref) / abs(ref):.1e}"]
    for k in T.NAMES:
        a, b = ggot[k], gref[k]
        err = (a - b).abs()
        mx = float(b.abs().max())
        over = int((err > 2e-5 * mx + 1e-4 * b.abs()).sum())
        line.append(f"{k}: max err {float(err.max()) / mx:.1e} of max, beyond {over}/{err.numel()}")
    print("; ".join(line))
