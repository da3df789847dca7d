#!/bin/bash
# the default bench line of the current tree
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06j
( time python bench.py ) > gpurun_out/r06j/bench_default.txt 2>&1
grep '^{"metric' gpurun_out/r06j/bench_default.txt | tail -1 > gpurun_out/r06j/bench_default.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06j/bench_default.json').read())
print('value', d['value'], 'ms', d['ms_per_step'])
print('deblur', {k:v for k,v in d['deblur'].items() if 'ms' in k or k=='unchanged_caller'})
print('flow', {k:v for k,v in d['get_flow'].items() if 'ms' in k})
print('train', {k:v for k,v in d['train_iteration'].items() if 'ms' in k or k=='unchanged_caller'})
print('repeat', d['repeat']['ms_per_step'], d['repeat']['forward_only_no_grad']['ms_per_render'])
print('dyn3', d['dynamic_config3']['ms_per_step'])
print('roofline', {k:v for k,v in d['roofline'].items() if k in ('achieved','frac','avg_kernel_ms','kernel')})
PY
tail -3 gpurun_out/r06j/bench_default.txt | cut -c1-300
