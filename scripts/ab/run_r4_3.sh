for v in "$@"; do
  echo "=== $v"; MOBGS_LIB=$GRAFT_REPO_ROOT/scripts/ab/lib$v.so timeout 600 scripts/prof_stall.sh stall_$v python $GRAFT_REPO_ROOT/scripts/check_bwd_mfma.py 300000 2>&1 | cut -c1-900
done
