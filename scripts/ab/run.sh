#!/bin/bash
# A/B timing of library builds on one box: scripts/ab/run.sh A B [A B ...]
for v in "$@"; do
  echo -n "$v: "; MOBGS_LIB=scripts/ab/lib$v.so python bench.py --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
done
