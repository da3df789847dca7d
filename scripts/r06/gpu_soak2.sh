#!/bin/bash
# second soak set of the final tree: full-size operator cases with long lists, the matrix-pipe arms, the side kernels
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06_soak2
( timeout 2400 python scripts/soak_parity.py --large --cases 24 ) > gpurun_out/r06_soak2/operator_large_24.txt 2>&1; tail -2 gpurun_out/r06_soak2/operator_large_24.txt
( timeout 1500 python scripts/soak_parity.py --no-heavy --cases 150 ) > gpurun_out/r06_soak2/operator_no_heavy_150.txt 2>&1; tail -2 gpurun_out/r06_soak2/operator_no_heavy_150.txt
( timeout 1500 python scripts/soak_misc.py ) > gpurun_out/r06_soak2/misc.txt 2>&1; tail -3 gpurun_out/r06_soak2/misc.txt
( timeout 1500 python scripts/soak_render.py --headline --flow --cases 30 ) > gpurun_out/r06_soak2/flow_30_headline_selection.txt 2>&1; tail -2 gpurun_out/r06_soak2/flow_30_headline_selection.txt
