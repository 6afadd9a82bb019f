from torch_random_walk_amd.rw import *  # noqa: F401,F403
from torch_random_walk_amd.rw import walk, walk_ex, set_rng_mode, get_rng_mode, set_check  # noqa: F401
from torch_random_walk_amd.rw import walk_triples, walk_triples_ex  # noqa: F401
from torch_random_walk_amd.rw import walk_edge_list, walk_edge_list_ex  # noqa: F401
from torch_random_walk_amd.rw import to_windows, to_windows_cbow, to_windows_ex, to_windows_cbow_ex  # noqa: F401
from torch_random_walk_amd.rw import to_windows_triples, to_windows_triples_cbow, to_windows_triples_ex  # noqa: F401
