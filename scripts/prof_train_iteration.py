"""Kernel profile driver of ONE whole training iteration at the headline size (scripts/prof.sh <name> python this)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
torch.autograd.set_multithreading_enabled(False)
import train_deblur_synth as TD
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tr = TD.DeblurTrainer("cuda:0", 200_000, 100_000, 1352, 1014, 2, iters=10000)
for _ in range(n):
    tr.iteration()
torch.cuda.synchronize()
