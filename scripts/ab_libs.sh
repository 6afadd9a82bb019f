#!/bin/bash
# Runs ON THE GPU BOX: scripts/tune_biased.py for several builds of the library on the same box, interleaved
# (LIBS: space-separated paths; default: the in-tree build and variants/*.so).  usage: ab_libs.sh [tune_biased args]
LIBS=${LIBS:-"torch_random_walk_amd/libtrw_hip.so $(ls variants/*.so 2>/dev/null)"}
for rep in 1 2; do
  for lib in $LIBS; do
    timeout -k 10 300 python scripts/tune_biased.py --lib $lib "$@" 2>&1 | grep -v amdgpu.ids
  done
done
