#!/bin/bash
# re-entry baseline of the current tree: whole GPU suite, smoke, default bench line, lean-step kernel profile
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06f
( time timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 ) > gpurun_out/r06f/gpu_suite.txt 2>&1
tail -25 gpurun_out/r06f/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py ) > gpurun_out/r06f/bench_default.txt 2>&1
tail -4 gpurun_out/r06f/bench_default.txt | cut -c1-1500
scripts/gpu_quick.sh r06f_lean > gpurun_out/r06f/lean.txt 2>&1; tail -30 gpurun_out/r06f/lean.txt
