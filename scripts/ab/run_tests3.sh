timeout 1800 python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_rccl_world1.py tests/test_gpu_bench_multirank.py tests/test_gpu_flow_loss.py -m gpu -q --tb=short 2>&1 | grep -v Warning | tail -25
python scripts/rccl_world1_check.py 2>&1 | tail -12
