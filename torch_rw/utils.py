from torch_random_walk_amd.utils import to_csr, nodes_tensor  # noqa: F401
from torch_random_walk_amd.utils import build_relation_tail_index  # noqa: F401
from torch_random_walk_amd.utils import to_edge_list_indexed, build_node_edge_index  # noqa: F401
