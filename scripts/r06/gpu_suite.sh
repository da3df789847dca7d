#!/bin/bash
# the whole GPU suite as the driver runs it (+ durations), then the smoke test
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06_suite
( time timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=25 ) > gpurun_out/r06_suite/gpu_suite.txt 2>&1
tail -45 gpurun_out/r06_suite/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
