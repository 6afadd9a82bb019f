"""Synthetic CSR graphs for the parity tests and bench.py (SURVEY 8(d) recipes).

Every generator is seeded, returns int64 ``(row_ptr, col_idx)`` with rows sorted
ascending (duplicates kept) and guarantees out-degree >= 1 by adding the ring
edge ``v -> (v+1) mod n`` -- the reference divides by zero on a degree-0 node
(csrc/cpu/rw_cpu.cpp:20).  Works on any torch device.
"""
import torch


def csr_from_edges(src, dst, n, add_ring=True):
    """Directed edge list -> CSR, rows sorted by destination."""
    dev = src.device
    if add_ring:
        ring = torch.arange(n, dtype=torch.int64, device=dev)
        src = torch.cat([src, ring])
        dst = torch.cat([dst, (ring + 1) % n])
    key = src * n + dst  # n < 2^31.5 keeps this inside int64
    key, _ = torch.sort(key)
    src_sorted = torch.div(key, n, rounding_mode="floor")
    col_idx = key - src_sorted * n
    counts = torch.bincount(src_sorted, minlength=n)
    row_ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=row_ptr[1:])
    return row_ptr, col_idx.contiguous()


def erdos_renyi(n, m, seed=1, device="cpu"):
    """m iid uniform directed edges on n nodes (+ ring)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    src = torch.randint(0, n, (m,), generator=g, device=device, dtype=torch.int64)
    dst = torch.randint(0, n, (m,), generator=g, device=device, dtype=torch.int64)
    return csr_from_edges(src, dst, n)


def rmat(n, m, seed=1, abcd=(0.57, 0.19, 0.19, 0.05), scale=None, device="cpu", chunk=1 << 25):
    """R-MAT power-law edges generated at 2^scale and folded ``mod n`` (+ ring)."""
    if scale is None:
        scale = max(1, (n - 1).bit_length())
    a, b, c, _ = abcd
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    srcs, dsts = [], []
    for lo in range(0, m, chunk):
        k = min(chunk, m - lo)
        s = torch.zeros(k, dtype=torch.int64, device=device)
        d = torch.zeros(k, dtype=torch.int64, device=device)
        for _ in range(scale):
            u = torch.rand(k, generator=g, device=device)
            sb = (u >= a + b).to(torch.int64)                      # quadrants c,d: source bit
            db = (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)  # quadrants b,d: dest bit
            s = (s << 1) | sb
            d = (d << 1) | db
        srcs.append(s % n)
        dsts.append(d % n)
    return csr_from_edges(torch.cat(srcs), torch.cat(dsts), n)


def shuffle_rows(row_ptr, col_idx, seed=0):
    """Same graph with every row's entries in a random order (exercises the
    unsorted-row membership scan)."""
    n = row_ptr.numel() - 1
    g = torch.Generator(device=col_idx.device)
    g.manual_seed(seed)
    deg = row_ptr[1:] - row_ptr[:-1]
    row_of = torch.repeat_interleave(torch.arange(n, device=col_idx.device), deg)
    noise = torch.rand(col_idx.numel(), generator=g, device=col_idx.device, dtype=torch.float64)
    order = torch.argsort(row_of.to(torch.float64) + noise * 0.999)
    return row_ptr.clone(), col_idx[order].contiguous()


def walks_follow_edges(row_ptr, col_idx, walks):
    """True iff every consecutive pair of every walk is an edge of the graph
    (a size-independent property of any correct walker)."""
    n = row_ptr.numel() - 1
    deg = row_ptr[1:] - row_ptr[:-1]
    row_of = torch.repeat_interleave(torch.arange(n, device=col_idx.device), deg)
    keys, _ = torch.sort(row_of * n + col_idx)
    ok = True
    L1 = walks.shape[1]
    for s in range(L1 - 1):
        q = walks[:, s] * n + walks[:, s + 1]
        pos = torch.searchsorted(keys, q).clamp_(max=keys.numel() - 1)
        ok = ok and bool((keys[pos] == q).all())
    return ok


def synthetic_kg(n_entities, n_triples, n_relations, seed=1, device="cpu", skew=(0.57, 0.19, 0.19, 0.05)):
    """Synthetic knowledge graph (BASELINE config 5 recipe): heads R-MAT-skewed, tails
    from the same generator's column bits, relations uniform.  Entity ids are 0..n_entities-1, relation
    ids n_entities..n_entities+n_relations-1 (the reference's tests number them after the entities,
    tests/test_rw_triples.py:14-24) and the padding id is n_entities+n_relations.
    Returns ``(triples [T,3] int64 unsorted, entities [n] int64, padding_idx)``.
    """
    scale = max(1, (n_entities - 1).bit_length())
    a, b, c, _ = skew
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    h = torch.zeros(n_triples, dtype=torch.int64, device=device)
    t = torch.zeros(n_triples, dtype=torch.int64, device=device)
    for _ in range(scale):
        u = torch.rand(n_triples, generator=g, device=device)
        h = (h << 1) | (u >= a + b).to(torch.int64)
        t = (t << 1) | (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)
    rel = torch.randint(0, n_relations, (n_triples,), generator=g, device=device, dtype=torch.int64) + n_entities
    triples = torch.stack([h % n_entities, rel, t % n_entities], dim=1).contiguous()
    entities = torch.arange(n_entities, dtype=torch.int64, device=device)
    return triples, entities, n_entities + n_relations
