timeout 600 scripts/prof.sh small_train_r4 python $GRAFT_REPO_ROOT/scripts/prof_small_train.py > /dev/null 2>&1
python - <<'PY'
import csv, os
rows=list(csv.DictReader(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/small_train_r4/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print("device ms per iteration:", tot/20/1e6, "launches per iteration", calls/20)
mob=sum(float(r['TotalDurationNs']) for r in rows if 'mobgs' in r['Name']); print("of which mobgs kernels:", mob/20/1e6, "launches", sum(int(r['Calls']) for r in rows if 'mobgs' in r['Name'])/20)
for r in rows[:22]:
    print(f"{r['Name'][:64]:64s} {int(r['Calls'])/20:7.1f}/it avg {float(r['AverageNs'])/1000:7.1f} us  per-it {float(r['TotalDurationNs'])/20/1000:8.1f} us")
PY
timeout 600 python scripts/bench_small_scene_iteration.py 2>&1 | tail -4
