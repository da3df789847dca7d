#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/prof.sh <name> <python script + args...>
# runs rocprofv3 --kernel-trace --stats and prints the per-kernel summary; output under gpurun_out/<name>/
set -u
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$name" -- "$@" > "$out/stdout.log" 2>&1
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
if [ -z "$f" ]; then echo "no kernel_stats.csv"; tail -20 "$out/stdout.log"; exit 1; fi
cp "$f" "$out/kernel_stats.csv"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
print(rows[0])
for r in rows[1:30]:
    print([c[:70] for c in r])
PY
# keep only the summaries (the raw trace can be large)
find "$out" -name '*kernel_trace.csv' -size +20M -delete
