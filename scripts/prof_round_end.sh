set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
python bench.py > "$out/bench_line.json" 2> "$out/bench_stderr.log"
head -c 400 "$out/bench_line.json"; echo
scripts/prof.sh ${tag}_bench python $root/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-cpu-torch --deblur-steps 0 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --small-steps 0 --repeat-steps 0 --no-kernel-breakdown > "$out/prof_bench.log" 2>&1
cp $root/gpurun_out/${tag}_bench/kernel_stats.csv "$out/bench_kernel_stats.csv"
scripts/prof.sh ${tag}_deblur python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cpu-torch --deblur-steps 6 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --small-steps 0 --repeat-steps 0 --no-kernel-breakdown > "$out/prof_deblur.log" 2>&1
cp $root/gpurun_out/${tag}_deblur/kernel_stats.csv "$out/deblur_kernel_stats.csv"
scripts/prof.sh ${tag}_flow python $root/scripts/prof_flow.py > "$out/prof_flow.log" 2>&1
cp $root/gpurun_out/${tag}_flow/kernel_stats.csv "$out/flow_kernel_stats.csv"
scripts/prof.sh ${tag}_small python $root/scripts/prof_small_scene.py --steps 100 --no-profile > "$out/prof_small.log" 2>&1
cp $root/gpurun_out/${tag}_small/kernel_stats.csv "$out/small_scene_kernel_stats.csv"
tail -3 $root/gpurun_out/${tag}_small/stdout.log
scripts/prof_pmc.sh ${tag}_pmc python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-cpu-torch --deblur-steps 0 --dynamic-steps 0 --flow-steps 0 --train-steps 0 --small-steps 0 --repeat-steps 0 --no-kernel-breakdown > "$out/pmc.log" 2>&1
cp $root/gpurun_out/${tag}_pmc/pmc_summary.json "$out/bench_pmc_summary.json"
tail -4 "$out/pmc.log"
