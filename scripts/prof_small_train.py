import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
torch.autograd.set_multithreading_enabled(False)
import train_deblur_synth as TD
tr = TD.DeblurTrainer("cuda:0", 20_000, 10_000, 512, 288, 2, iters=10000)
for _ in range(20):
    tr.iteration()
torch.cuda.synchronize()
