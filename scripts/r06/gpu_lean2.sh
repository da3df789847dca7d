#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r06n}
cd $root; mkdir -p gpurun_out/$tag
for rep in 1 2; do
scripts/gpu_quick.sh ${tag}_lean$rep > gpurun_out/$tag/lean$rep.txt 2>&1
python - gpurun_out/${tag}_lean$rep/kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls']) >= 60 and float(r['AverageNs']) > 4000:
        print("   %-60s %5s %8.1f" % (r['Name'].split('(')[0].replace('void ','').replace('mobgs::','')[:60], r['Calls'], float(r['AverageNs'])/1000))
PY
tail -3 gpurun_out/$tag/lean$rep.txt
done
