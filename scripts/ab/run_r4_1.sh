set -x
cd scripts/ubench && ./mfma_coissue > $GRAFT_REPO_ROOT/gpurun_out/mfma_coissue.txt 2>&1; cd $GRAFT_REPO_ROOT
tail -30 gpurun_out/mfma_coissue.txt
for v in M3 M2; do
  echo "=== $v full"; MOBGS_LIB=scripts/ab/lib$v.so timeout 300 python scripts/check_bwd_mfma.py 300000 2>&1 | tail -14
  echo "=== $v small"; MOBGS_LIB=scripts/ab/lib$v.so timeout 300 python scripts/check_bwd_mfma.py 30000 512 288 2>&1 | tail -14
done
