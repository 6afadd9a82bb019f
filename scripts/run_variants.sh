#!/bin/bash
# GPU box: time every prebuilt variants/*.so (scripts/build_variant.sh) with scripts/tune_biased.py.
# usage: scripts/run_variants.sh [names...]   env TUNE_ARGS: extra arguments (default: --what 14 --runs 2)
mkdir -p gpurun_out
names=("$@"); [ ${#names[@]} -eq 0 ] && names=($(ls variants/*.so | xargs -n1 basename | sed 's/\.so$//'))
for v in "${names[@]}"; do
  timeout -k 10 300 python3 ${TUNE_SCRIPT:-scripts/tune_biased.py} --lib variants/$v.so --tag "[$v]" ${TUNE_ARGS:---what 14 --runs 2} 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/variants.txt
done
