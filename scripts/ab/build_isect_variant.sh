#!/bin/bash
# library variant that differs only in isect.hip: scripts/ab/build_isect_variant.sh NAME "<-D flags>" -> scripts/ab/libNAME.so
# (links the other translation units' objects of the in-tree build: run `python -m mobgs_amd.build` first)
set -e
name=$1; flags=${2:-}
root=$(cd "$(dirname "$0")/../.." && pwd)
cd "$root/mobgs_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $flags -c isect.hip -o /tmp/isect_$name.o
objs=()
for f in *.hip; do
  [ $f = isect.hip ] && objs+=(/tmp/isect_$name.o) || objs+=(${f%.hip}.o)
done
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$root/scripts/ab/lib$name.so"
echo "built scripts/ab/lib$name.so"
