# per-kernel time of the headline bench for library variants: scripts/ab/kernel_ab.sh <kernel substring> A B ...
pat=$1; shift
for v in "$@"; do
  n=kab_${v}_$RANDOM
  MOBGS_LIB=$GRAFT_REPO_ROOT/scripts/ab/lib$v.so scripts/prof.sh $n python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-cpu-torch --deblur-steps 0 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --repeat-steps 0 --no-kernel-breakdown > /dev/null 2>&1
  python - $v $GRAFT_REPO_ROOT/gpurun_out/$n/kernel_stats.csv "$pat" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if sys.argv[3] in r['Name']:
        print(sys.argv[1], r['Name'][:40], r['Calls'], r['AverageNs'])
PY
done
