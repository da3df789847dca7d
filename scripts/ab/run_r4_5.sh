for v in "$@"; do
  echo "=== $v full"; MOBGS_LIB=$GRAFT_REPO_ROOT/scripts/ab/lib$v.so timeout 300 python $GRAFT_REPO_ROOT/scripts/check_bwd_mfma.py 300000 2>&1 | grep "WORST\|bwd_mfma=\|timing\|rror"
done
