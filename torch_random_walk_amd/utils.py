"""Graph -> tensor helpers used by the reference's README/tests for the CSR path
(torch_rw/utils.py:5-18), kept to the same names and results."""
import torch


def to_csr(graph):
    """networkx graph -> (row_ptr, col_idx) int64 tensors, rows sorted ascending.

    Reference: torch_rw/utils.py:5-9 (``nx.to_scipy_sparse_matrix(graph, format='csr')``,
    node order = ``graph.nodes()`` insertion order).  networkx >= 3 renamed that
    function; both spellings are accepted.
    """
    import networkx as nx
    conv = getattr(nx, "to_scipy_sparse_array", None) or getattr(nx, "to_scipy_sparse_matrix")
    csr = conv(graph, format="csr")
    csr.sort_indices()
    row_ptr = torch.as_tensor(csr.indptr, dtype=torch.int64).contiguous()
    col_idx = torch.as_tensor(csr.indices, dtype=torch.int64).contiguous()
    return row_ptr, col_idx


def nodes_tensor(graph):
    """``[0 .. n)`` as an int64 tensor (torch_rw/utils.py:11-18: the index of every
    node in ``graph.nodes()`` order, which is its position)."""
    return torch.arange(graph.number_of_nodes(), dtype=torch.int64)
