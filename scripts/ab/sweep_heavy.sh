# wall time per render() fwd+bwd for the default heavy-tile policy (-1) against none (0): bash scripts/ab/sweep_heavy.sh
for cfg in "512 288 20000 10000" "640 360 30000 15000" "800 448 40000 20000" "960 540 60000 30000"; do
  set -- $cfg
  for h in -1 0; do
    r=$(python scripts/prof_small_scene.py --width $1 --height $2 --ns $3 --nd $4 --steps 300 --no-profile --heavy $h 2>&1 | grep "fwd+bwd" | cut -d' ' -f2)
    echo "$1x$2 N=$(($3+$4)) heavy=$h: $r ms/step"
  done
done
