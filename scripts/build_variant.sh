#!/bin/bash
# HERE (no GPU needed): build a libtrw_hip.so variant with extra -D flags into variants/<name>.so; the directory is
# git-ignored but travels to the GPU box, where scripts/tune_biased.py --lib variants/<name>.so times it.
# usage: scripts/build_variant.sh <name> "<flags>"
set -e
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function $2 -o variants/$1.so \
    torch_random_walk_amd/csrc/trw_hip.hip torch_random_walk_amd/csrc/trw_cpu.cpp
echo "built variants/$1.so [$2]"
