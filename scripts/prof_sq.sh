#!/bin/bash
# usage: scripts/prof_sq.sh <name> <cmd...>   -- SQ counters (one pass, 8 slots) per kernel
set -u
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out" -o sq -- "$@" > "$out/log.txt" 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(f"{out}/**/*counter_collection.csv", recursive=True)
if not f:
    print(open(f"{out}/log.txt").read()[-2000:]); sys.exit(1)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"].split("(")[0][:48]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
    n = max(cnt[k], 1)
    print(k, "calls", n, {c: round(x / n) for c, x in v.items()})
PY
find "$out" -name '*.csv' -size +30M -delete
