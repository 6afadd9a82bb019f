"""Top-level alias so that `import torch_rw_native` (reference torch_rw/rw.py:1) resolves
to the MI355X extension built in torch_random_walk_amd/."""
from torch_random_walk_amd._native import ext as _ext

walk = _ext.walk
walk_ex = _ext.walk_ex
walk_triples = _ext.walk_triples
to_windows = _ext.to_windows
to_windows_cbow = _ext.to_windows_cbow
to_windows_triples = _ext.to_windows_triples
to_windows_triples_cbow = _ext.to_windows_triples_cbow
walk_edge_list = _ext.walk_edge_list
walk_edge_list_ex = _ext.walk_edge_list_ex
walk_triples_ex = _ext.walk_triples_ex
set_rng_mode = _ext.set_rng_mode
get_rng_mode = _ext.get_rng_mode
set_check = _ext.set_check
get_check = _ext.get_check
abi_version = _ext.abi_version
RNG_AUTO, RNG_GLIBC, RNG_WALKER = _ext.RNG_AUTO, _ext.RNG_GLIBC, _ext.RNG_WALKER
