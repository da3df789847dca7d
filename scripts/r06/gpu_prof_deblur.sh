#!/bin/bash
# per-kernel times of the deblur leg alone (K = 9 two-view iteration) and of the get_flow leg alone
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r06k}
cd $root; mkdir -p gpurun_out/$tag
for leg in deblur flow; do
  if [ $leg = deblur ]; then args="--deblur-steps 10 --flow-steps 0"; else args="--deblur-steps 0 --flow-steps 6"; fi
  scripts/prof.sh ${tag}_$leg python $root/bench.py --steps 2 --warmup 1 --prewarm 1 --no-cpu-baseline --no-cpu-torch $args --dynamic-steps 0 --train-steps 0 --small-steps 0 --repeat-steps 0 --no-kernel-breakdown > gpurun_out/$tag/prof_$leg.txt 2>&1
  echo "== $leg"
  python - gpurun_out/${tag}_$leg/kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:32]:
    print(f"  {r['Name'].split('(')[0].replace('void ','').replace('mobgs::','')[:64]:64s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1000:9.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
  grep '^{"metric' gpurun_out/${tag}_$leg/stdout.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('deblur', (d.get('deblur') or {}).get('ms_per_iteration'), 'unchanged', ((d.get('deblur') or {}).get('unchanged_caller') or {}).get('ms_per_iteration'), 'flow', {k:v for k,v in (d.get('get_flow') or {}).items() if 'ms' in k})"
done
