#!/usr/bin/env python
"""Per-step GPU timeline from a rocprofv3 kernel trace: busy time, gaps, kernel sequence.
    python scripts/timeline.py gpurun_out/<name>/<name>_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]) for r in rows)
starts = [i for i, e in enumerate(ev) if "prep_fwd_kernel" in e[2]]
i0, i1 = starts[-6], starts[-5]
seg = ev[i0:i1]
busy = sum(e[1] - e[0] for e in seg)
period = (ev[i1][0] - ev[i0][0]) / 1e3
print(f"kernels/step {len(seg)}  period {period:.1f} us  busy {busy / 1e3:.1f} us  idle {period - busy / 1e3:.1f} us")
prev = seg[0][1]
for s, e, n in seg:
    gap = (s - prev) / 1e3 if s > prev else 0.0
    print(f"{(s - seg[0][0]) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {n}")
    prev = max(prev, e)
