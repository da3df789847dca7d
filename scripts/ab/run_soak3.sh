# second set of seeds on the final kernels: full-size cases, the quadrant forward kernel forced, render paths
mkdir -p gpurun_out/soak_r4c
run() { out=gpurun_out/soak_r4c/$1.txt; shift; timeout 1500 "$@" > $out 2>&1; tail -1 $out; }
run operator_large_40 python scripts/soak_parity.py --large --cases 40 --seed 4301
run operator_default_300 python scripts/soak_parity.py --cases 300 --seed 4302
MOBGS_BWD_MFMA=2 run operator_arm2_large_16 python scripts/soak_parity.py --large --cases 16 --seed 4303
run render_default_80 python scripts/soak_render.py --cases 80 --seed 4304
run render_flow_40 python scripts/soak_render.py --flow --cases 40 --seed 4305
