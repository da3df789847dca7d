bash scripts/prof_round_end.sh r04z 2>&1 | tail -12
python bench.py --steps 20 --warmup 5 > gpurun_out/r04z/bench_line_driver_invocation.json 2> /dev/null; head -c 300 gpurun_out/r04z/bench_line_driver_invocation.json; echo
timeout 600 python scripts/bench_small_scene_iteration.py > gpurun_out/r04z/small_scene_iteration.txt 2>&1; tail -3 gpurun_out/r04z/small_scene_iteration.txt
timeout 300 python scripts/bench_graphed.py > gpurun_out/r04z/bench_graphed.txt 2>&1; tail -4 gpurun_out/r04z/bench_graphed.txt
