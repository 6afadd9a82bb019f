#!/bin/bash
# GPU box: build libtrw_hip.so variants with extra -D flags and time each (scripts/tune_biased.py).
# usage: scripts/tune_variants.sh "<flags of variant 1>" "<flags of variant 2>" ...   (use "" for the stock build)
# env TUNE_ARGS: extra arguments for tune_biased.py
set -u
mkdir -p gpurun_out/variants
i=0
for FLAGS in "$@"; do
  i=$((i+1))
  LIB=gpurun_out/variants/libtrw_v$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function $FLAGS -o $LIB \
      torch_random_walk_amd/csrc/trw_hip.hip torch_random_walk_amd/csrc/trw_cpu.cpp 2> gpurun_out/variants/build_v$i.log || { echo "build failed: $FLAGS"; tail -5 gpurun_out/variants/build_v$i.log; continue; }
  timeout -k 10 300 python3 ${TUNE_SCRIPT:-scripts/tune_biased.py} --lib $LIB --tag "[$FLAGS]" ${TUNE_ARGS:-} 2>&1 | grep -v amdgpu.ids
  rm -f $LIB
done
