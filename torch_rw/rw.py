from torch_random_walk_amd.rw import *  # noqa: F401,F403
from torch_random_walk_amd.rw import walk, walk_ex, set_rng_mode, get_rng_mode, set_check  # noqa: F401
