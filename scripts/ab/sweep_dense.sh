# bin_kernel time with the LDS-ranked (dense) binning forced on against the default: bash scripts/ab/sweep_dense.sh
for cfg in "704 400 34000 17000" "960 540 60000 30000" "1352 1014 200000 100000"; do
  set -- $cfg
  for f in "" "--dense"; do
    n=sd_$1_${f#--}
    r=$(scripts/prof.sh $n python $GRAFT_REPO_ROOT/scripts/prof_small_scene.py --width $1 --height $2 --ns $3 --nd $4 --steps 60 --no-profile $f | grep "bin_kernel" | cut -d, -f4)
    echo "$1x$2 $f: bin avg ns $r"
  done
done
