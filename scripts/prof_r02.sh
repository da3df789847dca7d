#!/bin/bash
# Round-2 evidence in one GPU call: scripts/prof_r02.sh <tag>   (run from the repo root on the GPU box)
#   kernel-trace stats of bench.py (lean + deblur legs) and of scripts/bench_deform.py, PMC traffic + SQ counters of the
#   compositors, the FETCH_SIZE / WRITE_SIZE calibration of scripts/ubench/fetch_calib.  Summaries land in
#   gpurun_out/<tag>/ ; copy what is to be judged into profiles/r02/.
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
python bench.py --steps 50 --warmup 10 > "$out/bench_line.json" 2> "$out/bench_stderr.log"
tail -c 600 "$out/bench_line.json"; echo
scripts/prof.sh ${tag}_bench python $root/bench.py --steps 60 --warmup 10 --no-cpu-baseline --deblur-steps 0 > "$out/prof_bench.log" 2>&1
cp $root/gpurun_out/${tag}_bench/kernel_stats.csv "$out/bench_kernel_stats.csv" 2>/dev/null
tail -1 $root/gpurun_out/${tag}_bench/stdout.log > "$out/bench_line_under_rocprof.json"
scripts/prof.sh ${tag}_deblur python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --deblur-steps 6 > "$out/prof_deblur.log" 2>&1
cp $root/gpurun_out/${tag}_deblur/kernel_stats.csv "$out/deblur_kernel_stats.csv" 2>/dev/null
scripts/prof.sh ${tag}_deform python $root/scripts/bench_deform.py --n 100000 --steps 20 > "$out/prof_deform.log" 2>&1
cp $root/gpurun_out/${tag}_deform/kernel_stats.csv "$out/deform_kernel_stats.csv" 2>/dev/null
python scripts/bench_deform.py > "$out/bench_deform.json" 2>/dev/null
scripts/prof_pmc.sh ${tag}_pmc python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --deblur-steps 0 > "$out/pmc.log" 2>&1
cp $root/gpurun_out/${tag}_pmc/pmc_summary.json "$out/bench_pmc_summary.json" 2>/dev/null
scripts/prof_sq.sh ${tag}_sq python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --deblur-steps 0 > "$out/sq_counters.txt" 2>&1
scripts/prof_pmc.sh ${tag}_calib $root/scripts/ubench/fetch_calib > "$out/calib.log" 2>&1
cp $root/gpurun_out/${tag}_calib/pmc_summary.json "$out/fetch_calib_pmc_summary.json" 2>/dev/null
head -12 "$out/bench_kernel_stats.csv" | cut -c1-150
cat "$out/sq_counters.txt" | tail -8 | cut -c1-400
cat "$out/calib.log" | tail -8
cat "$out/pmc.log" | tail -6
