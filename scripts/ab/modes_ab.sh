# deblur iteration + small scene for library variants (train-mode renders go through raster_layers / class passes)
for v in "$@"; do
  r=$(MOBGS_LIB=$GRAFT_REPO_ROOT/scripts/ab/lib$v.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-cpu-torch --deblur-steps 10 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --repeat-steps 0 --no-kernel-breakdown 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['deblur']['ms_per_iteration'], d['deblur']['unchanged_caller']['ms_per_iteration'])")
  echo "$v: lean ms, deblur ms, unchanged deblur ms = $r"
done
