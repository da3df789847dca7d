#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06t
timeout 900 python -X faulthandler -m pytest tests/test_gpu_graphed.py -x -q -m gpu 2>&1 | grep -v "^  \|Warning\|^$" | tail -8
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06t/bench.txt 2>&1
grep '^{"metric' gpurun_out/r06t/bench.txt | tail -1 > gpurun_out/r06t/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06t/bench.json').read())
print('value', d['value'], 'ms', d['ms_per_step'])
print('train', {k:v for k,v in d['train_iteration'].items() if 'ms' in k or k=='graphed'})
print('small', d.get('reference_operating_point'))
print('roofline', {k:v for k,v in d['roofline'].items() if k in ('achieved','frac','avg_kernel_ms','kernel','algorithmic_bytes_formula')})
PY
tail -4 gpurun_out/r06t/bench.txt | cut -c1-200
