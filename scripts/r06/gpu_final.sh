#!/bin/bash
# round-end set of the final tree (one GPU call): suite, smoke, PMC + bench lines, per-kernel profiles of every leg, SQ counters,
# the reference's operating point, secondary modes.  Everything lands under gpurun_out/r06z/ (copied to profiles/r06/ by hand).
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; out=gpurun_out/r06z; mkdir -p $out
( time timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=25 ) > $out/gpu_suite.txt 2>&1
grep -E "passed|failed" $out/gpu_suite.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.txt
# the driver's invocation with --pmc: collects FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU for these kernel sources
python bench.py --pmc --steps 20 --warmup 5 > $out/bench_line_pmc.json 2> $out/bench_pmc_stderr.log
cp profiles/r06_raster_bwd_pmc.json $out/r06_raster_bwd_pmc.json 2>/dev/null
head -c 300 $out/bench_line_pmc.json; echo
python bench.py > $out/bench_line.json 2> $out/bench_stderr.log
head -c 300 $out/bench_line.json; echo
scripts/prof.sh r06z_bench python $root/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-cpu-torch --deblur-steps 0 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --small-steps 0 --repeat-steps 0 --no-kernel-breakdown > $out/prof_bench.log 2>&1
cp gpurun_out/r06z_bench/kernel_stats.csv $out/bench_kernel_stats.csv
scripts/r06/gpu_prof_deblur.sh r06zl > $out/prof_legs.log 2>&1
cp gpurun_out/r06zl_deblur/kernel_stats.csv $out/deblur_kernel_stats.csv; cp gpurun_out/r06zl_flow/kernel_stats.csv $out/flow_kernel_stats.csv
scripts/prof.sh r06z_small python $root/scripts/prof_small_scene.py --steps 100 --no-profile > $out/prof_small.log 2>&1
cp gpurun_out/r06z_small/kernel_stats.csv $out/small_scene_kernel_stats.csv
scripts/prof_sq.sh r06z_sq python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-cpu-torch --deblur-steps 0 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --small-steps 0 --repeat-steps 0 --no-kernel-breakdown > $out/sq_counters.txt 2>&1
python scripts/bench_small_scene_iteration.py > $out/small_scene_iteration.txt 2>&1
python scripts/bench_graphed.py > $out/bench_graphed.txt 2>&1
python scripts/r06/graph_small_iteration.py > $out/graph_small_iteration.txt 2>&1
LAMBDA_FLOW=0 python scripts/r06/graph_small_iteration.py >> $out/graph_small_iteration.txt 2>&1
python scripts/bench_modes.py > $out/bench_modes.json 2> $out/bench_modes_stderr.log
tail -3 $out/graph_small_iteration.txt | cut -c1-250
