for mode in "--zero-weight" "--zero-weight --separate --no-sink"; do
n=fz_$RANDOM
scripts/prof.sh $n python $GRAFT_REPO_ROOT/scripts/prof_flow.py $mode > /dev/null 2>&1
echo "=== $mode"; grep "ms" gpurun_out/$n/stdout.log | tail -2
python - $GRAFT_REPO_ROOT/gpurun_out/$n/kernel_stats.csv <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("device ms per step:", round(tot/7e6,3))
for r in rows[:24]:
    print(f"{r['Name'][:66]:66s} {int(r['Calls'])/7:6.1f}/step avg {float(r['AverageNs'])/1000:8.1f} us  per-step {float(r['TotalDurationNs'])/7/1000:8.1f}")
PY
done
