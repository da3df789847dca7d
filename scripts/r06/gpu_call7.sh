#!/bin/bash
# decoder-backward prologue with the streaming row reduction: parity tests + lean-step kernel profile
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06h
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_fused_decode_bwd.py tests/test_gpu_fused_decode.py tests/test_gpu_render_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graphed.py -x -v -m gpu > gpurun_out/r06h/pytest.log 2>&1
grep -n "PASSED\|FAILED\|ERROR\|Fatal\|fault\|passed\|failed" gpurun_out/r06h/pytest.log | tail -30
scripts/gpu_quick.sh r06h_lean > gpurun_out/r06h/lean.txt 2>&1
tail -4 gpurun_out/r06h/lean.txt
python - gpurun_out/r06h_lean/kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls']) >= 60:
        print("   %-60s %5s %8.1f" % (r['Name'].split('(')[0].replace('void ','').replace('mobgs::','')[:60], r['Calls'], float(r['AverageNs'])/1000))
PY
