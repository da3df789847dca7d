export MOBGS_ARMS=0,3
for v in "$@"; do
  echo "=== $v full"; MOBGS_LIB=$GRAFT_REPO_ROOT/scripts/ab/lib$v.so timeout 300 scripts/prof.sh chk_$v python $GRAFT_REPO_ROOT/scripts/check_bwd_mfma.py 300000 2>&1 | grep "raster_bwd\|slot_reduce\|Fill" | cut -c1-200
  grep "arm 3\|WORST\|bwd_mfma=\|rror" gpurun_out/chk_$v/stdout.log
done
