#!/bin/bash
# like build_variant.sh, but the extra flags go to raster.hip ONLY (WITHOUT the in-tree build's own scheduler flag) and every other object is reused from the in-tree build
# (mobgs_amd/csrc/*.o): scripts/ab/build_variant_raster.sh NAME "<flags>"  -> scripts/ab/libNAME.so
set -e
name=$1; flags=${2:-}
root=$(cd "$(dirname "$0")/../.." && pwd)
cd "$root/mobgs_amd/csrc"
tmp=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize $flags -c raster.hip -o $tmp/raster.o
objs=()
for f in *.hip; do o=${f%.hip}.o; if [ "$f" = raster.hip ]; then objs+=($tmp/raster.o); else objs+=($o); fi; done
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$root/scripts/ab/lib$name.so"
rm -rf $tmp
echo "built scripts/ab/lib$name.so"
