# decoder kernels per library variant: scripts/ab/kernel_ab_dec.sh A B ...
for v in "$@"; do
  n=kab_${v}_$RANDOM
  MOBGS_LIB=$GRAFT_REPO_ROOT/scripts/ab/lib$v.so scripts/prof.sh $n python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-cpu-torch --deblur-steps 0 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --repeat-steps 0 --no-kernel-breakdown > /dev/null 2>&1
  python - $v $GRAFT_REPO_ROOT/gpurun_out/$n/kernel_stats.csv <<'PY'
import csv, sys
out = [sys.argv[1]]
for r in csv.DictReader(open(sys.argv[2])):
    if 'decoder' in r['Name']:
        out.append(f"{r['Name'][7:34]} {float(r['AverageNs'])/1000:.1f}")
print(" | ".join(out))
PY
done
