"""Which torch (non-mobgs) device kernels does one view's nine get_flow() calls launch?  (the glue)"""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from torch.profiler import profile, ProfilerActivity
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda")
scam, cam, stat, dyn, raw = B.build_scene(dev, 200_000, 100_000, 1352, 1014)
fw = B.FlowWorkload(dev, stat, dyn, cam, 1352, 1014, zero=False)
for _ in range(2):
    fw.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    fw.step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    if e.device_time_total > 0 and not e.key.startswith("mobgs") and "mobgs::" not in e.key:
        rows.append((e.device_time_total, e.count, e.key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("non-mobgs device time per view: %.2f ms" % (tot / 1e3))
for t, c, k in rows[:40]:
    print("%8.1f us %5d  %s" % (t, c, k[:110]))
