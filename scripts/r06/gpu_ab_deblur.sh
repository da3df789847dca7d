#!/bin/bash
# A/B of the deblur leg: decoder-backward prologue and cover_slots on / off
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06l
for v in "MOBGS_FUSE_DECODER_BWD=0 MOBGS_COVER_SLOTS=0" "MOBGS_FUSE_DECODER_BWD=1 MOBGS_COVER_SLOTS=0" "MOBGS_FUSE_DECODER_BWD=0 MOBGS_COVER_SLOTS=1" "MOBGS_FUSE_DECODER_BWD=1 MOBGS_COVER_SLOTS=1" "MOBGS_FUSE_DECODER_BWD=0 MOBGS_COVER_SLOTS=0" "MOBGS_FUSE_DECODER_BWD=1 MOBGS_COVER_SLOTS=1"; do
  env $v python bench.py --steps 5 --warmup 2 --prewarm 2 --no-cpu-baseline --no-cpu-torch --deblur-steps 20 --flow-steps 0 --dynamic-steps 0 --train-steps 0 --small-steps 0 --repeat-steps 0 --no-kernel-breakdown 2>/dev/null | grep '^{"metric' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('$v', 'deblur', d['deblur']['ms_per_iteration'], 'median', d['deblur']['event_median_ms_per_iteration'], 'unchanged', d['deblur']['unchanged_caller']['ms_per_iteration'])"
done
