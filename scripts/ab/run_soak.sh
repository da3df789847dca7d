# randomised soaks with the matrix-pipe backward: default policy (small grids -> team kernel), and both arms forced
mkdir -p gpurun_out/soak_r4
timeout 1500 python scripts/soak_parity.py --cases 300 --seed 4101 > gpurun_out/soak_r4/operator_default_300.txt 2>&1; tail -3 gpurun_out/soak_r4/operator_default_300.txt
MOBGS_BWD_MFMA=1 timeout 1500 python scripts/soak_parity.py --cases 300 --seed 4102 --no-heavy > gpurun_out/soak_r4/operator_arm1_noheavy_300.txt 2>&1; tail -3 gpurun_out/soak_r4/operator_arm1_noheavy_300.txt
MOBGS_BWD_MFMA=2 timeout 1500 python scripts/soak_parity.py --cases 300 --seed 4103 > gpurun_out/soak_r4/operator_arm2_300.txt 2>&1; tail -3 gpurun_out/soak_r4/operator_arm2_300.txt
MOBGS_BWD_MFMA=1 timeout 1500 python scripts/soak_parity.py --large --cases 24 --seed 4104 > gpurun_out/soak_r4/operator_arm1_large_24.txt 2>&1; tail -3 gpurun_out/soak_r4/operator_arm1_large_24.txt
MOBGS_BWD_MFMA=2 timeout 1500 python scripts/soak_parity.py --large --cases 24 --seed 4105 > gpurun_out/soak_r4/operator_arm2_large_24.txt 2>&1; tail -3 gpurun_out/soak_r4/operator_arm2_large_24.txt
timeout 1500 python scripts/soak_render.py --cases 120 --seed 4106 > gpurun_out/soak_r4/render_default_120.txt 2>&1; tail -3 gpurun_out/soak_r4/render_default_120.txt
MOBGS_BWD_MFMA=2 timeout 1500 python scripts/soak_render.py --cases 80 --seed 4107 > gpurun_out/soak_r4/render_arm2_80.txt 2>&1; tail -3 gpurun_out/soak_r4/render_arm2_80.txt
timeout 1500 python scripts/soak_render.py --many --cases 40 --seed 4108 > gpurun_out/soak_r4/render_many_default_40.txt 2>&1; tail -3 gpurun_out/soak_r4/render_many_default_40.txt
