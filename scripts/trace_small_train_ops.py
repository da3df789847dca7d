"""Which torch (non-mobgs) device kernels does ONE training iteration at 512x288 / 30 k splats launch, by aten op?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import train_deblur_synth as TD
from torch.profiler import profile, ProfilerActivity
torch.autograd.set_multithreading_enabled(False)
tr = TD.DeblurTrainer("cuda:0", 20_000, 10_000, 512, 288, 2, iters=10000)
for _ in range(5):
    tr.iteration()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.iteration()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    if e.device_time_total > 0 and "mobgs::" not in e.key and not e.key.startswith("void "):
        rows.append((e.self_device_time_total, e.device_time_total, e.count, e.key))
rows.sort(reverse=True)
print("self device us | total device us | calls | op")
for st, t, c, k in rows[:60]:
    print("%8.1f %8.1f %5d  %s" % (st, t, c, k[:100]))
