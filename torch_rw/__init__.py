"""Drop-in name of the reference package: `from torch_rw import rw, utils`."""
from torch_random_walk_amd import rw, utils  # noqa: F401
