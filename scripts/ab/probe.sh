#!/bin/bash
# kernel time of bin_kernel for library variants: scripts/ab/probe.sh A R S ...
for v in "$@"; do
  MOBGS_LIB=$GRAFT_REPO_ROOT/scripts/ab/lib$v.so timeout 100 scripts/prof.sh probe_$v python $GRAFT_REPO_ROOT/scripts/bin_probe.py 2>&1 | grep "bin_kernel\|emit_kernel\|tile_scan\|Error\|error" | cut -c1-140 | sed "s/^/$v: /"
done
