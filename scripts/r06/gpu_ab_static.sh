#!/bin/bash
# A/B of MobgsTuning.static_rows on one box: per-kernel microseconds of the lean step, arms interleaved
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out/r06b
if [ $# -gt 0 ]; then timeout 1200 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -8; fi
for arm in 1 0 1 0; do
  MOBGS_STATIC_ROWS=$arm scripts/gpu_quick.sh r06b_static${arm} > gpurun_out/r06b/ab_static${arm}_$RANDOM.txt 2>&1
  echo "static_rows=$arm: $(grep -h 'raster_bwd_kernel<10' gpurun_out/r06b_static$arm/kernel_stats.csv | awk -F, '{print $(NF-6)}' | head -1) $(tail -3 gpurun_out/r06b/ab_static${arm}_*.txt | tr '\n' ' ')"
  python - gpurun_out/r06b_static$arm/kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'raster_bwd_kernel' in r['Name'] or 'raster_fwd_blocks' in r['Name']:
        print("   ", r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1000)
PY
done
