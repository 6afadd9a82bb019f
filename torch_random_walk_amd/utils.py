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


def build_relation_tail_index(triples_indexed_tensor, all_entities_tensor):
    """Sort the triples by head and index them: returns ``(relation_tail_index, triples_sorted)``.

    Reference: torch_rw/utils.py:91-120 -- a pandas sort by head followed by a per-triple Python loop that
    records, for every entity id, the inclusive (first, last) position of its triples, ``(-1, -1)`` when
    it heads none.  This is the same result from one stable sort and two scatters (the order of
    triples that share a head is their input order, which is what the reference's golden walks
    assume, tests/test_rw_triples.py:70-79); works on any device.
    """
    triples = triples_indexed_tensor.to(torch.int64)
    num_nodes = int(all_entities_tensor.numel())
    heads = triples[:, 0].contiguous()
    order = torch.sort(heads, stable=True).indices
    triples_sorted = triples[order].contiguous()
    heads = triples_sorted[:, 0]
    num_edges = int(heads.numel())
    index = torch.full((num_nodes, 2), -1, dtype=torch.int64, device=triples.device)
    if num_edges == 0:
        return index, triples_sorted
    pos = torch.arange(num_edges, dtype=torch.int64, device=triples.device)
    is_first = torch.ones(num_edges, dtype=torch.bool, device=triples.device)
    is_first[1:] = heads[1:] != heads[:-1]
    is_last = torch.ones(num_edges, dtype=torch.bool, device=triples.device)
    is_last[:-1] = heads[1:] != heads[:-1]
    index[heads[is_first], 0] = pos[is_first]
    index[heads[is_last], 1] = pos[is_last]
    if num_edges == 1:
        index[heads[0], 1] = -1  # the reference's loop never closes the range of a lone triple (utils.py:106-118)
    return index, triples_sorted


def to_edge_list_indexed(graph):
    """networkx graph -> (edge_list_indexed [E,2] int64, node -> index dict).

    Reference: torch_rw/utils.py:21-56.  Node indices are positions in ``sorted(graph.nodes())``;
    the dict is filled in the order nodes are met while walking ``graph.edges()`` (head before
    tail), which is the order the reference's tests take their ``target_nodes`` from; an
    undirected graph gets every edge in both directions (the flipped copies appended).
    """
    import networkx as nx
    edges = list(graph.edges())
    position = {node: i for i, node in enumerate(sorted(graph.nodes()))}
    node_index_mapping = {}
    pairs = []
    for head, tail in edges:
        for node in (head, tail):
            if node not in node_index_mapping:
                node_index_mapping[node] = position[node]
        pairs.append((position[head], position[tail]))
    edge_list_indexed = torch.tensor(pairs, dtype=torch.int64).reshape(-1, 2)
    if not nx.is_directed(graph):
        edge_list_indexed = torch.cat((edge_list_indexed, torch.fliplr(edge_list_indexed)), dim=0)
    return edge_list_indexed.contiguous(), node_index_mapping


def build_node_edge_index(edge_list_indexed, nodes_tensor):
    """Sort the edges by head and index them: returns ``(node_edge_index [n,2], edges_sorted)``.

    Reference: torch_rw/utils.py:58-89 (pandas sort by head + per-edge Python loop): for every node
    id the inclusive (first, last) position of its out-edges, ``(-1, -1)`` when it has none.  Same
    construction as ``build_relation_tail_index``: one stable sort, two scatters, any device.
    """
    edges = edge_list_indexed.to(torch.int64)
    num_nodes = int(torch.unique(nodes_tensor).numel())
    order = torch.sort(edges[:, 0].contiguous(), stable=True).indices
    edges_sorted = edges[order].contiguous()
    heads = edges_sorted[:, 0]
    num_edges = int(heads.numel())
    index = torch.full((num_nodes, 2), -1, dtype=torch.int64, device=edges.device)
    if num_edges == 0:
        return index, edges_sorted
    pos = torch.arange(num_edges, dtype=torch.int64, device=edges.device)
    boundary = heads[1:] != heads[:-1]
    is_first = torch.ones(num_edges, dtype=torch.bool, device=edges.device)
    is_first[1:] = boundary
    is_last = torch.ones(num_edges, dtype=torch.bool, device=edges.device)
    is_last[:-1] = boundary
    index[heads[is_first], 0] = pos[is_first]
    index[heads[is_last], 1] = pos[is_last]
    if num_edges == 1:
        index[heads[0], 1] = -1  # the reference's loop never closes the range of a lone edge (utils.py:75-87)
    return index, edges_sorted
