#!/bin/bash
# build a library variant for A/B timing on one box: scripts/ab/build_variant.sh NAME "<extra hipcc flags for the
# compositing translation units>"  -> scripts/ab/libNAME.so   (run here, on the CPU container: hipcc cross-compiles)
set -e
name=$1; flags=${2:-}
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d)
cd "$root/mobgs_amd/csrc"
objs=()
for f in *.hip; do
  extra=""
  case $f in raster.hip) extra="-fno-slp-vectorize -mllvm -misched-prera-direction=topdown $flags";; raster_bwd_mfma.hip|raster_layers.hip) extra="-fno-slp-vectorize $flags";; project.hip) extra="-ffp-contract=off $flags";; *) extra="$flags";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $extra -c $f -o $tmp/${f%.hip}.o &
  objs+=($tmp/${f%.hip}.o)
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$root/scripts/ab/lib$name.so"
rm -rf $tmp
echo "built scripts/ab/lib$name.so"
