# raw geometric sums in the gradient slots (conic applied in stage 2): parity first, then the per-kernel times
timeout 1500 python -m pytest tests/test_gpu_operator_parity.py tests/test_gpu_render_parity.py tests/test_gpu_bwd_mfma.py tests/test_gpu_bruteforce.py tests/test_gpu_fullsize.py tests/test_gpu_edge_cases.py tests/test_gpu_known_answers.py tests/test_gpu_flow_loss.py -m gpu -x -q 2>&1 | grep -v Warning | tail -12
n=raw_$RANDOM
scripts/prof.sh $n python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-cpu-torch --deblur-steps 0 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --repeat-steps 0 --no-kernel-breakdown > /dev/null 2>&1
python - $GRAFT_REPO_ROOT/gpurun_out/$n/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'raster' in r['Name'] or 'slot_reduce' in r['Name']:
        print(f"{r['Name'][:70]:70s} {float(r['AverageNs'])/1000:.1f}")
PY
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-cpu-torch --no-kernel-breakdown 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step')}); print('deblur',d.get('deblur',{}).get('ms_per_step'),'flow',d.get('get_flow',{}).get('ms_per_view'),'train',d.get('train_iteration',{}).get('ms_per_iteration'))"
