#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06d
timeout 1500 python -m pytest tests/test_gpu_clamp_flag.py tests/test_gpu_static_rows.py tests/test_gpu_zero_gate.py tests/test_gpu_optim.py "tests/test_gpu_fullsize.py::test_fullsize_lean_render_against_the_oracle_chain" tests/test_gpu_fused_lists.py tests/test_gpu_bruteforce.py tests/test_gpu_operator_parity.py -x -q -m gpu 2>&1 | tail -12
scripts/ab.sh kernels main plaineval main plaineval 2>&1 | tee gpurun_out/r06d/ab_plaineval.txt
python scripts/r06/train_iter_probe.py 12 2 > gpurun_out/r06d/train_probe.txt 2>&1; tail -16 gpurun_out/r06d/train_probe.txt
