#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06h
for v in "MOBGS_FUSE_DECODER_BWD=0 MOBGS_FASTPATH=0" "MOBGS_FUSE_DECODER_BWD=1 MOBGS_FASTPATH=0"; do
  echo "== $v"
  env $v timeout 600 python -X faulthandler -m pytest tests/test_gpu_graphed.py tests/test_gpu_fused_decode_bwd.py -x -q -m gpu 2>&1 | grep -v "^  File\|^$" | tail -30
done
timeout 900 python -m pytest tests/test_gpu_fused_decode_bwd.py -x -q -m gpu 2>&1 | tail -3
scripts/gpu_quick.sh r06h_lean > gpurun_out/r06h/lean.txt 2>&1
python - gpurun_out/r06h_lean/kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls']) >= 60:
        print("   %-60s %5s %8.1f" % (r['Name'].split('(')[0].replace('void ','').replace('mobgs::','')[:60], r['Calls'], float(r['AverageNs'])/1000))
PY
tail -3 gpurun_out/r06h/lean.txt
