from torch_random_walk_amd.utils import to_csr, nodes_tensor  # noqa: F401
