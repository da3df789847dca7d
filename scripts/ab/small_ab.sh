# kernel times of the 512x288 / 30 k scene for library variants: scripts/ab/small_ab.sh A B
for v in "$@"; do
  n=small_${v}_$RANDOM
  MOBGS_LIB=$GRAFT_REPO_ROOT/scripts/ab/lib$v.so scripts/prof.sh $n python $GRAFT_REPO_ROOT/scripts/prof_small_scene.py --steps 100 --no-profile > /dev/null 2>&1
  python - $v $GRAFT_REPO_ROOT/gpurun_out/$n/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if 'bin_kernel' in r['Name'] or 'scan_lookback' in r['Name']:
        print(sys.argv[1], r['Name'][:34], r['Calls'], r['AverageNs'])
PY
  grep "ms per step\|ms/step" $GRAFT_REPO_ROOT/gpurun_out/$n/stdout.log | tail -1
done
