#!/bin/bash
# A/B experiments with library variants (MOBGS_LIB): one parametrised script instead of a directory of one-off ones.
#
#   scripts/ab.sh build NAME [FILE.hip|all] "<extra hipcc flags>"    (CPU container: hipcc cross-compiles)
#        -> scripts/ab/libNAME.so; FILE.hip: only that translation unit is recompiled with the extra flags (on top of
#           the in-tree build's own per-file flags), the other objects come from the in-tree build; all: every unit
#   scripts/ab.sh kernels V1 V2 ... [-- PATTERN]    (GPU box) per-kernel microseconds of the lean step for each variant
#        ("main" = the in-tree library); PATTERN: regex of kernel names to print (default: the compositors)
#   scripts/ab.sh lists V1 V2 ...                   (GPU box) per-kernel microseconds of projection -> tile lists alone
#        (scripts/lists_probe.py; MOBGS_PROBE_HINT / MOBGS_PROBE_SPATIAL are honoured)
#   scripts/ab.sh step V1 V2 [V1 V2 ...]            (GPU box) wall-clock ms per lean step, every run and per-variant means
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cmd=$1; shift
lib_of() { [ "$1" = main ] && echo "$root/mobgs_amd/csrc/libmobgs_hip.so" || echo "$root/scripts/ab/lib$1.so"; }
BENCH_LEAN="--no-cpu-baseline --no-cpu-torch --deblur-steps 0 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --small-steps 0 --repeat-steps 0 --no-kernel-breakdown"
case $cmd in
build)
  name=$1; unit=${2:-all}; flags=${3:-}
  mkdir -p "$root/scripts/ab"; cd "$root/mobgs_amd/csrc"; tmp=$(mktemp -d); objs=()
  for f in *.hip; do
    own=""
    case $f in raster.hip) own="-fno-slp-vectorize -mllvm -misched-prera-direction=topdown";; raster_bwd_mfma.hip|raster_layers.hip) own="-fno-slp-vectorize";; project.hip) own="-ffp-contract=off";; esac
    if [ "$unit" = all ] || [ "$unit" = "$f" ]; then
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $own $flags -c $f -o $tmp/${f%.hip}.o &
      objs+=($tmp/${f%.hip}.o)
    else
      objs+=(${f%.hip}.o)   # (python -m mobgs_amd.build first)
    fi
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$root/scripts/ab/lib$name.so"; rm -rf $tmp
  echo "built scripts/ab/lib$name.so";;
kernels)
  vs=(); pat='raster_bwd_kernel|raster_fwd'
  while [ $# -gt 0 ]; do if [ "$1" = -- ]; then pat=$2; shift 2; else vs+=("$1"); shift; fi; done
  for v in "${vs[@]}"; do
    n=kab_${v}_$RANDOM
    MOBGS_LIB=$(lib_of $v) "$root/scripts/prof.sh" $n python "$root/bench.py" --steps 60 --warmup 10 $BENCH_LEAN > /dev/null 2>&1 || true
    python - "$v" "$root/gpurun_out/$n/kernel_stats.csv" "$pat" <<'PY'
import csv, re, sys
out, tot = [sys.argv[1]], 0.0
for r in csv.DictReader(open(sys.argv[2])):
    if int(r['Calls']) >= 60:
        tot += float(r['TotalDurationNs']) / int(r['Calls']) / 1000
    if re.search(sys.argv[3], r['Name']):
        out.append(f"{r['Name'].split('(')[0].replace('void mobgs::', '')[:34]} {float(r['AverageNs'])/1000:.1f}")
print(" | ".join(out), f"| all per-step kernels {tot:.1f} us")
PY
  done;;
lists)
  for v in "$@"; do
    MOBGS_LIB=$(lib_of $v) timeout 200 "$root/scripts/prof.sh" lab_$v python "$root/scripts/lists_probe.py" > /dev/null 2>&1 || true
    python - "$v" "$root/gpurun_out/lab_$v/kernel_stats.csv" <<'PY'
import csv, sys
out, tot = [sys.argv[1]], 0.0
for r in csv.DictReader(open(sys.argv[2])):
    n = r['Name']
    for k in ('project_fwd', 'scan_lookback', 'bin_kernel', 'tile_finish', 'tile_sort_seg', 'tile_scan', 'emit_kernel', 'tile_sort_short'):
        if k in n and int(r['Calls']) >= 25:
            out.append(f"{k} {float(r['AverageNs'])/1000:.1f}")
            tot += float(r['AverageNs'])/1000
print(" | ".join(out), f"| sum {tot:.1f}")
PY
    grep -h '^I ' "$root/gpurun_out/lab_$v/stdout.log" | cut -c1-100
  done;;
step)
  declare -A sum cnt
  for v in "$@"; do
    r=$(MOBGS_LIB=$(lib_of $v) python "$root/bench.py" --steps 300 --warmup 30 $BENCH_LEAN 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])")
    echo "$v: $r"
    sum[$v]=$(python -c "print(${sum[$v]:-0}+$r)"); cnt[$v]=$(( ${cnt[$v]:-0} + 1 ))
  done
  for v in "${!sum[@]}"; do python -c "print('mean $v: %.4f ms over ${cnt[$v]} runs' % (${sum[$v]}/${cnt[$v]}))"; done;;
*) sed -n 2,14p "$0"; exit 1;;
esac
