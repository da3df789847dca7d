#!/bin/bash
# usage (on the GPU box): scripts/gpu_quick.sh <tag> [pytest args...]
# runs the given GPU tests (if any), then the lean-step kernel profile; prints the per-kernel table
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
if [ $# -gt 0 ]; then
  timeout 900 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -15
fi
scripts/prof.sh $tag python $root/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-cpu-torch --deblur-steps 0 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --small-steps 0 --repeat-steps 0 --no-kernel-breakdown | cut -c1-150 | head -24
python - $root/gpurun_out/$tag/kernel_stats.csv <<'PY'
import csv, sys
tot = 0.0; comp = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls']) >= 60:
        us = float(r['TotalDurationNs']) / int(r["Calls"]) / 1000
        tot += us
        if 'raster_bwd_kernel' in r['Name'] or 'raster_fwd' in r['Name']:
            comp += us
print(f"sum of per-step kernels {tot:.1f} us, compositors {comp:.1f} us, everything else {tot - comp:.1f} us")
PY
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $root/gpurun_out/$tag/stdout.log | head -2
