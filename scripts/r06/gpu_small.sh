#!/bin/bash
# the reference's own operating point (512x288, 30 k splats): eager lean step with the round-6 switches on / off, graphed step,
# whole small-scene iteration
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/r06s
for v in "MOBGS_FUSE_DECODER_BWD=0 MOBGS_COVER_SLOTS=0" "MOBGS_FUSE_DECODER_BWD=1 MOBGS_COVER_SLOTS=1" "MOBGS_FUSE_DECODER_BWD=0 MOBGS_COVER_SLOTS=0" "MOBGS_FUSE_DECODER_BWD=1 MOBGS_COVER_SLOTS=1"; do
  echo "$v: $(env $v python scripts/bench_small_eager.py 2>/dev/null | tail -1)"
done
python scripts/bench_graphed.py 2>/dev/null | tail -4
python scripts/bench_small_scene_iteration.py 2>/dev/null | tail -8
