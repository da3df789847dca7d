timeout 1800 python -m pytest tests/test_gpu_flow_loss.py tests/test_gpu_graphed.py -m gpu -q --tb=short 2>&1 | grep -v Warning | tail -12
timeout 1500 python bench.py --steps 20 --warmup 5 --pmc > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err; tail -c 6000 gpurun_out/r4_bench1.json; tail -5 gpurun_out/r4_bench1.err
cp profiles/r04_raster_bwd_pmc.json gpurun_out/ 2>/dev/null
