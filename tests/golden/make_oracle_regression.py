#!/usr/bin/env python3
"""Regression vectors of the ORACLE itself: checksums of its walks on seeded graphs,
so that a change to oracle/trw_oracle.c that alters any walk is caught even where the reference has no golden.
The oracle is pinned on the reference's goldens and on live libc elsewhere (tests/test_oracle.py); this file
only freezes what it produced at the time those pins were green.

    python tests/golden/make_oracle_regression.py > tests/golden/oracle_regression.json
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O  # noqa: E402
from tests.helpers import hub_graph, random_graph  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int64).tobytes()).hexdigest()


def cases():
    rp, ci = random_graph(200, 4, 7)                       # item 3: 200-node graph, N=300, L=20
    tg = np.random.default_rng(3).integers(0, 200, 300)
    for seed in (0, 1, 10, 123456, -1):
        yield "g200_uniform_seed%d" % seed, lambda s=seed: O.walk(rp, ci, tg, 1.0, 1.0, 20, s)
    for p, q in ((0.25, 4.0), (0.7, 0.5), (2.0, 1.0), (1.0, 1.0 + 1e-12)):
        yield "g200_biased_glibc_p%g_q%g" % (p, q), lambda p=p, q=q: O.walk(rp, ci, tg, p, q, 20, 10)
        yield "g200_biased_walker_p%g_q%g" % (p, q), \
            lambda p=p, q=q: O.walk(rp, ci, tg, p, q, 20, 10, rng_mode=O.RNG_WALKER, walker_base=5)
    hp, hc = hub_graph(1024, 9)                            # item 4: hub graph, N=512, L=80
    htg = np.random.default_rng(4).integers(0, 1024, 512)
    yield "hub_uniform", lambda: O.walk(hp, hc, htg, 1.0, 1.0, 80, 10)
    yield "hub_biased_walker", lambda: O.walk(hp, hc, htg, 0.25, 4.0, 80, 10, rng_mode=O.RNG_WALKER)
    yield "hub_biased_glibc", lambda: O.walk(hp, hc, htg[:64], 0.25, 4.0, 80, 10)
    yield "L1", lambda: O.walk(rp, ci, tg, 0.5, 2.0, 1, 3)          # item 5: edge cases
    yield "duplicates", lambda: O.walk(rp, ci, np.full(50, 7), 0.5, 2.0, 10, 3)
    yield "indexed_far", lambda: O.walk(rp, ci, tg[:32], 1.0, 1.0, 80, 10, rng_mode=O.RNG_GLIBC_INDEXED,
                                        walker_base=2**33 // 80 + 99)


if __name__ == "__main__":
    print(json.dumps({name: digest(fn()) for name, fn in cases()}, indent=1, sort_keys=True))
