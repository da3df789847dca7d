# randomised soaks after the raw-sum gradient slots: new seeds, default policy + both matrix-pipe arms + flow + large
mkdir -p gpurun_out/soak_r4b
run() { out=gpurun_out/soak_r4b/$1.txt; shift; timeout 1500 "$@" > $out 2>&1; tail -2 $out; }
run operator_default_300 python scripts/soak_parity.py --cases 300 --seed 4201
run operator_noheavy_200 python scripts/soak_parity.py --cases 200 --seed 4202 --no-heavy
MOBGS_BWD_MFMA=1 run operator_arm1_noheavy_200 python scripts/soak_parity.py --cases 200 --seed 4203 --no-heavy
MOBGS_BWD_MFMA=2 run operator_arm2_200 python scripts/soak_parity.py --cases 200 --seed 4204
run operator_large_24 python scripts/soak_parity.py --large --cases 24 --seed 4205
run operator_bwd_blocks_100 python scripts/soak_parity.py --cases 100 --seed 4206 --bwd-blocks
run render_default_120 python scripts/soak_render.py --cases 120 --seed 4207
run render_many_40 python scripts/soak_render.py --many --cases 40 --seed 4208
run render_flow_40 python scripts/soak_render.py --flow --cases 40 --seed 4209
