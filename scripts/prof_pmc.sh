#!/bin/bash
# usage (GPU box, repo root): scripts/prof_pmc.sh <name> <cmd...>
# Two separate counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950: TCC has 4 slots,
# MI355X_MICROARCH.md "rocprofv3 PMC slots"); no tracing flags are combined with --pmc.
set -u
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d "$out/$ctr" -o "$ctr" -- "$@" > "$out/$ctr.log" 2>&1
done
python - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{out}/{ctr}/**/*counter_collection.csv", recursive=True)
    if not files:
        print("no counter file for", ctr); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != ctr:
                continue
            k = row["Kernel_Name"].split("(")[0]
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    for k, (tot, n) in acc.items():
        res[k][ctr] = tot / n; res[k]["calls"] = n
rows = sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))
json.dump({k: v for k, v in rows}, open(f"{out}/pmc_summary.json", "w"), indent=1)
for k, v in rows[:16]:
    print(f"{k[:60]:60s} calls={v.get('calls')} FETCH_SIZE/launch={v.get('FETCH_SIZE',0):.0f} WRITE_SIZE/launch={v.get('WRITE_SIZE',0):.0f}")
PY
find "$out" -name '*.csv' -size +30M -delete
