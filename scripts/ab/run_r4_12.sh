for z in 1 0; do
  echo "=== MOBGS_PREZERO_SLOTS=$z"
  MOBGS_PREZERO_SLOTS=$z scripts/prof.sh pz_$z python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-cpu-torch --deblur-steps 0 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --repeat-steps 0 --no-kernel-breakdown 2>&1 | grep "raster_fwd\|Fill\|raster_bwd\|decoder_bwd" | cut -c1-160
done
