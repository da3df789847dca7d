# per-kernel time of the get_flow leg (wide compositor builds) for library variants
for v in "$@"; do
  n=fab_${v}_$RANDOM
  MOBGS_LIB=$GRAFT_REPO_ROOT/scripts/ab/lib$v.so scripts/prof.sh $n python $GRAFT_REPO_ROOT/scripts/prof_flow.py > /dev/null 2>&1
  python - $v $GRAFT_REPO_ROOT/gpurun_out/$n/kernel_stats.csv <<'PY'
import csv, sys
out = [sys.argv[1]]; tot = 0
for r in csv.DictReader(open(sys.argv[2])):
    tot += float(r['TotalDurationNs'])
    if 'raster_' in r['Name'] and 'pack' not in r['Name']:
        out.append(f"{r['Name'][12:40]} {float(r['AverageNs'])/1000:.0f}")
print(" | ".join(out), "| total ms/step", round(tot / 7e6, 3))
PY
done
