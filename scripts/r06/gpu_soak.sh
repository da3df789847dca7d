#!/bin/bash
# randomised soaks of the round-6 tree against the oracles (long forms; logs -> profiles/r06/soak/) + SQ counters of the lean step
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06_soak
( timeout 1500 python scripts/soak_render.py --cases 100 ) > gpurun_out/r06_soak/render_100.txt 2>&1; tail -3 gpurun_out/r06_soak/render_100.txt
( timeout 1500 python scripts/soak_render.py --many --cases 40 ) > gpurun_out/r06_soak/render_many_40.txt 2>&1; tail -3 gpurun_out/r06_soak/render_many_40.txt
( timeout 1500 python scripts/soak_render.py --flow --cases 30 ) > gpurun_out/r06_soak/flow_30.txt 2>&1; tail -3 gpurun_out/r06_soak/flow_30.txt
( timeout 1500 python scripts/soak_parity.py --cases 200 ) > gpurun_out/r06_soak/operator_200.txt 2>&1; tail -3 gpurun_out/r06_soak/operator_200.txt
# the headline kernel selection on the soak's small images too: one wave per tile, the quadrant backward (+ decoder prologue, cover)
( timeout 1500 python scripts/soak_render.py --headline --cases 100 ) > gpurun_out/r06_soak/render_100_headline_selection.txt 2>&1; tail -3 gpurun_out/r06_soak/render_100_headline_selection.txt
( timeout 1500 python scripts/soak_render.py --headline --many --cases 40 ) > gpurun_out/r06_soak/render_many_40_headline_selection.txt 2>&1; tail -3 gpurun_out/r06_soak/render_many_40_headline_selection.txt
scripts/prof_sq.sh r06_sq python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-cpu-torch --deblur-steps 0 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --small-steps 0 --repeat-steps 0 --no-kernel-breakdown > gpurun_out/r06_soak/sq_counters.txt 2>&1; tail -8 gpurun_out/r06_soak/sq_counters.txt
