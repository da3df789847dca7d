#!/bin/bash
# the GPU tests around the backward compositor (cover_slots, decoder prologue, operator / render parity, graphs, soaks), then the lean-step kernel profile
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06i
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_cover_slots.py tests/test_gpu_fused_decode_bwd.py tests/test_gpu_operator_parity.py tests/test_gpu_render_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graphed.py tests/test_gpu_static_rows.py tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r06i/pytest.log 2>&1
grep -v "^  \|Warning\|^$" gpurun_out/r06i/pytest.log | tail -12
scripts/gpu_quick.sh r06i_lean > gpurun_out/r06i/lean.txt 2>&1
python - gpurun_out/r06i_lean/kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls']) >= 60:
        print("   %-60s %5s %8.1f" % (r['Name'].split('(')[0].replace('void ','').replace('mobgs::','')[:60], r['Calls'], float(r['AverageNs'])/1000))
PY
tail -3 gpurun_out/r06i/lean.txt
