#!/bin/bash
# GPU box: the configs[3]-size prepare (111 M nodes / 1.6 G distinct entries asked of the generator) -> gpurun_out/r05_papers100m_prepare.txt:
# per-kernel times of the full prepare and of one shard's node2vec walk, the phase clocks of the eight-wave tile kernel, the
# call's wall time next to the kernels' (bench.py --workload papers100m_biased reports both as cold.*).
O=$GRAFT_REPO_ROOT/gpurun_out/papers_prep; mkdir -p $O
R=$GRAFT_REPO_ROOT/gpurun_out/r05_papers100m_prepare.txt
{
echo "# scripts/papers_prepare_r05.sh: rocprofv3 --kernel-trace --stats of scripts/tune_biased.py --nodes 111000000 --edges 1600000000 --walkers 13875000 --what 14 --runs 2"
bash $GRAFT_REPO_ROOT/scripts/prof_papers_prepare.sh
echo "# the same graph, -DTRW_TILE_CLOCKS=1 build (wave cycles by phase in the eight-wave tile kernel)"
cd $GRAFT_REPO_ROOT
python3 scripts/tune_biased.py --lib variants/clk1.so --nodes 111000000 --edges 1600000000 --walkers 13875000 --what 14 --runs 1 --tag clk1 2>&1 | grep -v amdgpu | tail -4
} > $R 2>&1
tail -30 $R
