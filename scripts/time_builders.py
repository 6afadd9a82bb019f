#!/usr/bin/env python3
"""Device time of the graph -> tensor builders (row f-4) at BASELINE scale; run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_random_walk_amd import graphs, utils  # noqa: E402

dev = "cuda:0"


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


triples, ents, pad = graphs.synthetic_kg(1_000_000, 5_000_000, 1000, seed=1, device=dev)
print("build_relation_tail_index  5 M triples / 1 M entities: %.2f ms on the GPU, %.0f ms on the host (same code)"
      % (timed(lambda: utils.build_relation_tail_index(triples, ents)),
         timed(lambda: utils.build_relation_tail_index(triples.cpu(), ents.cpu()), reps=1)))
rp, ci = graphs.rmat(1_000_000, 10_000_000, seed=1, device=dev)
src = torch.repeat_interleave(torch.arange(1_000_000, device=dev), rp[1:] - rp[:-1])
edges = torch.stack([src, ci], 1)[torch.randperm(ci.numel(), device=dev)]
nodes = torch.arange(1_000_000, device=dev)
print("build_node_edge_index      11 M edges / 1 M nodes:     %.2f ms on the GPU, %.0f ms on the host (same code)"
      % (timed(lambda: utils.build_node_edge_index(edges, nodes)),
         timed(lambda: utils.build_node_edge_index(edges.cpu(), nodes.cpu()), reps=1)))
