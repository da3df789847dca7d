for z in 1 0; do
  echo "=== MOBGS_PREZERO_SLOTS=$z"
  MOBGS_PREZERO_SLOTS=$z timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-cpu-torch --deblur-steps 5 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --repeat-steps 0 --no-kernel-breakdown 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['deblur']['ms_per_iteration'], d['deblur']['unchanged_caller']['ms_per_iteration'])"
done
timeout 1800 python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_operator_parity.py tests/test_gpu_fullsize.py tests/test_gpu_render_many.py tests/test_gpu_config4.py tests/test_gpu_graphed.py tests/test_gpu_train_loop.py -m gpu -q --tb=short 2>&1 | grep -v Warning | tail -8
