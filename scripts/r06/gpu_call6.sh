#!/bin/bash
# decoder backward as the prologue of the backward compositor: parity tests, then the lean-step kernel profile with it on / off
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06g
timeout 1500 python -m pytest tests/test_gpu_fused_decode_bwd.py tests/test_gpu_fused_decode.py tests/test_gpu_render_parity.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r06g/pytest.log
tail -25 gpurun_out/r06g/pytest.log
for arm in 1 0 1 0; do
  MOBGS_FUSE_DECODER_BWD=$arm scripts/gpu_quick.sh r06g_decb$arm > gpurun_out/r06g/ab_decb${arm}_$RANDOM.txt 2>&1
  echo "FUSE_DECODER_BWD=$arm: $(tail -3 gpurun_out/r06g/ab_decb${arm}_*.txt | tr '\n' ' ')"
  python - gpurun_out/r06g_decb$arm/kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('raster_bwd_kernel','raster_fwd_blocks','decoder_','slot_reduce')):
        print("   ", r['Name'].split('(')[0][-48:], r['Calls'], float(r['AverageNs'])/1000)
PY
done
