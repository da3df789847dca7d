#!/bin/bash
# A/B timing of library builds on one box: scripts/ab/run.sh A B [A B ...]; prints every run and the per-variant mean
declare -A sum cnt
for v in "$@"; do
  r=$(MOBGS_LIB=scripts/ab/lib$v.so python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-cpu-torch --deblur-steps 0 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --repeat-steps 0 --no-kernel-breakdown 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])")
  echo "$v: $r"
  ms=${r%% *}
  sum[$v]=$(python -c "print(${sum[$v]:-0}+$ms)"); cnt[$v]=$(( ${cnt[$v]:-0} + 1 ))
done
for v in "${!sum[@]}"; do python -c "print('mean $v: %.4f ms over ${cnt[$v]} runs' % (${sum[$v]}/${cnt[$v]}))"; done
