#!/bin/bash
# GPU box: scripts/check_prepare_repeat.py in N fresh processes of 3 prepares each (flips, when they come, come early)
N=${1:-10}; shift
for i in $(seq $N); do TRW_LIB=$TRW_LIB python scripts/check_prepare_repeat.py 10000000 100000000 3 14 2>&1 | grep -v amdgpu | grep -E "differ|entry" | grep -v " 0 entries" | sed "s/^/run $i: /"; done
echo "done $N runs"
