# GPU test-suite under the default policy and with the backward MFMA arms forced
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -15
for arm in 1 2; do
  echo "=== MOBGS_BWD_MFMA=$arm"
  MOBGS_BWD_MFMA=$arm timeout 1800 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_edge_cases.py tests/test_gpu_known_answers.py tests/test_gpu_render_many.py tests/test_gpu_config3.py tests/test_gpu_config4.py tests/test_gpu_soak.py tests/test_gpu_graphed.py tests/test_gpu_train_loop.py -m gpu -q 2>&1 | grep -v Warning | tail -8
done
