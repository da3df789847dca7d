#!/bin/bash
# per-kernel time of the list pipeline for library variants: scripts/ab/lists_ab.sh A B ...  ("main" = the in-tree build)
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/scripts/ab/lib$v.so; [ $v = main ] && lib=$GRAFT_REPO_ROOT/mobgs_amd/csrc/libmobgs_hip.so
  MOBGS_LIB=$lib timeout 200 scripts/prof.sh lab_$v python $GRAFT_REPO_ROOT/scripts/lists_probe.py > /dev/null 2>&1
  python - $v $GRAFT_REPO_ROOT/gpurun_out/lab_$v/kernel_stats.csv <<'PY'
import csv, sys
out = [sys.argv[1]]
tot = 0
for r in csv.DictReader(open(sys.argv[2])):
    n = r['Name']
    for k in ('project_fwd', 'scan_lookback', 'bin_kernel<true, true>', 'bin_kernel<false, true>', 'tile_finish', 'tile_sort_seg'):
        if k in n and int(r['Calls']) >= 25:
            out.append(f"{k.split('<')[0]} {float(r['AverageNs'])/1000:.1f}")
            tot += float(r['AverageNs'])/1000
print(" | ".join(out), f"| sum {tot:.1f}")
PY
  grep -h '^I ' $GRAFT_REPO_ROOT/gpurun_out/lab_$v/stdout.log | cut -c1-100
done
