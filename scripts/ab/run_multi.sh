for w in 2 8; do
MOBGS_BENCH_SHARE_GPU=1 MOBGS_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$w --master-addr 127.0.0.1 --master-port $((29500+w)) bench.py --gpus $w --steps 3 --warmup 1 --prewarm 2 > gpurun_out/bench_gpus${w}_gloo_shared_gpu.log 2>&1
grep "^{" gpurun_out/bench_gpus${w}_gloo_shared_gpu.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['n_gpus'], d['value'], d['scale_anchor'], d['scaling_efficiency_vs_anchor'], d['data'][:40])"
done
timeout 600 python -m pytest tests/test_gpu_bench_multirank.py -m gpu -q 2>&1 | tail -2
