#!/bin/bash
# bin<DENSE> for camera batches: list-equality tests, then the deblur / flow legs under rocprofv3 (per-kernel times)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06e
true
scripts/prof.sh r06e_deblur python $root/bench.py --steps 5 --warmup 2 --prewarm 2 --no-cpu-baseline --no-cpu-torch --deblur-steps 10 --dynamic-steps 0 --flow-steps 6 --train-steps 0 --repeat-steps 0 --no-kernel-breakdown > gpurun_out/r06e/prof_deblur.txt 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r06e_deblur/kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:22]:
    print(f"  {r['Name'].split('(')[0].replace('void ','').replace('mobgs::','')[:58]:58s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1000:9.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
tail -1 gpurun_out/r06e_deblur/stdout.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('lean', d['value'], 'deblur', d.get('deblur',{}).get('ms_per_iteration'), 'unchanged', d.get('deblur',{}).get('unchanged_caller',{}).get('ms_per_iteration'), 'flow', {k:v for k,v in (d.get('get_flow') or d.get('flows') or {}).items() if 'ms' in k})"
