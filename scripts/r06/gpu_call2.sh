#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06c
timeout 1200 python -m pytest tests/test_gpu_zero_gate.py "tests/test_gpu_fullsize.py::test_fullsize_lean_render_against_the_oracle_chain" tests/test_gpu_fused_prep.py -x -q -m gpu -s 2>&1 | tail -12
python scripts/r06/train_iter_probe.py 20 2 > gpurun_out/r06c/train_probe.txt 2>&1; tail -28 gpurun_out/r06c/train_probe.txt
python scripts/r06/train_iter_probe.py 12 2 unchanged > gpurun_out/r06c/train_probe_unchanged.txt 2>&1; tail -18 gpurun_out/r06c/train_probe_unchanged.txt
python scripts/soak_render.py --help 2>&1 | tail -5
