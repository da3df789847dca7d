#!/bin/bash
# round 6, GPU call 1: the new parity tests on the headline kernel selection + A/B of the static-row blend body
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out/r06a
timeout 1500 python -m pytest tests/test_gpu_static_rows.py tests/test_gpu_render_parity.py tests/test_gpu_fused_decode.py tests/test_gpu_fused_prep.py tests/test_gpu_config4.py::test_blurry_view_k9_blce_matches_reference_fixture "tests/test_gpu_fullsize.py::test_fullsize_lean_render_against_the_oracle_chain" -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r06a/pytest.log
tail -30 gpurun_out/r06a/pytest.log
for arm in 1 0 1 0; do
  MOBGS_STATIC_ROWS=$arm scripts/gpu_quick.sh r06a_static$arm > gpurun_out/r06a/ab_static${arm}_$RANDOM.txt 2>&1
  tail -3 gpurun_out/r06a/ab_static${arm}_*.txt | tail -3
  grep -h "raster_bwd_kernel\|raster_fwd_blocks" gpurun_out/r06a_static$arm/kernel_stats.csv | cut -c1-200 | head -3
done
