timeout 900 python -m pytest tests/test_gpu_config5.py::test_fp16_storage_training_tracks_fp32 -m gpu -q --tb=short 2>&1 | grep -v Warning | tail -12
MOBGS_BWD_MFMA=0 timeout 900 python -m pytest tests/test_gpu_config5.py::test_fp16_storage_training_tracks_fp32 -m gpu -q --tb=short 2>&1 | grep -v Warning | tail -12
for arm in 1 2; do
MOBGS_BWD_MFMA=$arm timeout 900 python -m pytest "tests/test_gpu_render_parity.py" tests/test_gpu_operator_parity.py -m gpu -q --tb=short 2>&1 | grep -v Warning | tail -20
done
