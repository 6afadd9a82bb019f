"""The reference's Python surface for the CSR walk (torch_rw/rw.py:3-4), on MI355X.

``walk`` keeps the reference's name, parameter names and order.  ``walk_ex`` adds
the two things a sharded caller needs (global walker base, explicit RNG mode).
"""
from ._native import ext as torch_rw_native
from ._native import RNG_AUTO, RNG_GLIBC, RNG_WALKER  # noqa: F401


def walk(row_ptr, col_idx, target_nodes, p, q, walk_length, seed):
    """Random walks of ``walk_length`` steps from every node in ``target_nodes``.

    Same contract as the reference (torch_rw/rw.py:3-4 -> csrc/rw_init.cpp:11-25):
    int64 CSR tensors in, ``[len(target_nodes), walk_length+1]`` int64 out on the
    inputs' device, column 0 = start node; ``p == 1.0 and q == 1.0`` selects the
    uniform walk, anything else the node2vec rejection sampler.

    Random streams (``RNG_AUTO``): uniform walks always reproduce ``csrc/cpu``'s libc stream.  A
    node2vec batch of at most ``32768 // (walk_length + 1)`` walkers (404 at L=80: the reference's ``parallel_for``
    grain, below which its output is defined) replays that ONE serial stream too; a larger batch runs the
    per-walker Philox stream -- so the same ``(seed, node)`` walks differently in a 404- and in a 405-walker call.
    ``walk_ex(rng_mode=RNG_WALKER)`` (or ``set_rng_mode``) makes small batches use the per-walker stream as well.
    """
    return torch_rw_native.walk(row_ptr, col_idx, target_nodes, p, q, walk_length, seed)


CHECK_OFF, CHECK_STRICT, CHECK_DEFERRED = (int(torch_rw_native.CHECK_OFF), int(torch_rw_native.CHECK_STRICT),
                                           int(torch_rw_native.CHECK_DEFERRED))


def _check_code(check):
    """None -> the process-wide mode (set_check); True -> strict (synchronise, raise now); False -> never
    look; "deferred" -> asynchronous read-back, raised by the next call or by check_status()."""
    if check is None:
        return -1
    if check is True:
        return CHECK_STRICT
    if check is False:
        return CHECK_OFF
    if check == "deferred" or check == CHECK_DEFERRED:
        return CHECK_DEFERRED
    raise ValueError("check must be None, True, False or 'deferred'")


def walk_ex(row_ptr, col_idx, target_nodes, p, q, walk_length, seed, walker_base=0, rng_mode=RNG_AUTO,
            check=None, out=None):
    """``walk`` with an explicit global index for ``target_nodes[0]`` and RNG mode.

    A shard that passes the global offset of its slice reproduces exactly the rows
    the unsharded call would have produced (include/trw.h, TRW_RNG_*).  ``out``: a contiguous
    int64 ``[len(target_nodes), walk_length + 1]`` tensor to walk into (e.g. the shard's slice of
    the gathered result) instead of a fresh one -- or an **int32** one for NARROW rows: the node ids as
    uint32 bit patterns (``widen_rows`` gives the int64 rows back), the wire format of the multi-GPU
    gather, written by the walk kernel itself where the production kernels run (the reference-exact
    uniform walk; the node2vec walk on the per-walker stream over a prepared graph) and by a narrowing
    pass behind an int64 walk elsewhere: the same values either way.  GPU tensors, node ids below 2^32.
    """
    return torch_rw_native.walk_ex(row_ptr, col_idx, target_nodes, p, q, walk_length, seed, walker_base,
                                   rng_mode, _check_code(check), out)


def narrow_rows(rows, out=None, strict=False):
    """The uint32 wire format of a block of int64 rows (multi-GPU gather, include/trw.h ``trw_rows_narrow_*``):
    an int32 tensor of the same shape holding the low 32 bits.  ``strict`` synchronises and raises when a value
    did not fit.  GPU tensors run the HIP kernel, CPU tensors the library's host leg."""
    return torch_rw_native.narrow_rows(rows, out, strict)


def widen_rows(wire, out=None):
    """Inverse of ``narrow_rows``: zero-extends the uint32 values of an int32 tensor into int64 rows."""
    return torch_rw_native.widen_rows(wire, out)


PREP_UNIFORM, PREP_BIASED, PREP_MARKS, PREP_MARKS_FULL = (
    int(torch_rw_native.PREP_UNIFORM), int(torch_rw_native.PREP_BIASED), int(torch_rw_native.PREP_MARKS),
    int(torch_rw_native.PREP_MARKS_FULL))


def prepare(row_ptr, col_idx, biased=True, uniform=False, marks=True):
    """Build now what the GPU walks derive from ``(row_ptr, col_idx)`` alone and keep it for every later
    ``walk`` over the same tensors (no reference counterpart: the reference re-reads the int64 CSR on each call).

    ``walk`` does this by itself -- the first call over a graph prepares what one call is worth, later node2vec
    calls add the neighbourhood masks of longer rows once enough steps have been made -- so ``prepare`` is only a way
    to pay up front: ``biased``: edge set + 32-byte fat entries (node2vec walks); ``marks``: with the neighbourhood
    masks of rows up to 256 entries (about 1.5x the time, 45 % faster walks), ``marks="full"``: up to 1024 entries
    (about 2x the time, another 20 %); ``uniform``: 16-byte fat entries for ``p == q == 1``.  (R-MAT 10M/110M: 38 / 53 /
    68 ms to prepare, 80 / 45 / 37 ms per 8e8-step walk; 56 / 64 / 64 bytes per edge.)
    Valid while the two tensors are alive and not written to (an in-place write re-prepares on the next walk).
    Returns the ``PREP_*`` bits the cached graph holds.
    """
    what = (PREP_BIASED if biased or marks else 0) | (PREP_MARKS if marks else 0) | (PREP_UNIFORM if uniform else 0)
    if marks == "full":
        what |= PREP_MARKS_FULL
    return torch_rw_native.prepare(row_ptr, col_idx, what)


def set_cache(on=True, capacity=0):
    """Switch the prepared-graph cache on/off (off: every call prepares into its own workspace);
    ``capacity`` > 0 sets how many graphs are kept (default 2, least recently used dropped first)."""
    torch_rw_native.set_cache(bool(on), int(capacity))


def invalidate(row_ptr, col_idx):
    """Forget what is cached for these two tensors (``(edge_list_indexed, node_edge_index)`` for the edge-list walk).
    The cache notices in-place writes through the tensors' version counters; a write torch cannot see -- through
    ``tensor.data``, DLPack / cupy views, another extension's kernels -- leaves them unchanged: call this after such a
    write (the reference re-reads the CSR on every call and needs nothing of the kind).  Returns the entries dropped."""
    return torch_rw_native.invalidate(row_ptr, col_idx)


def clear_cache():
    """Drop every cached prepared graph (their device memory goes back to the allocator)."""
    torch_rw_native.clear_cache()


def cache_info():
    """``[(num_nodes, nnz, PREP_* bits, bytes, node2vec walks served), ...]``, most recently used first."""
    return torch_rw_native.cache_info()


def prepare_timing():
    """Host-side cost of the last prepared-graph build of this process: ``{"alloc_ms": what the allocator took for
    the prepared graph and the marking passes' scratch, "launch_ms": what enqueueing the kernels took,
    "prepared_bytes", "scratch_bytes"}`` (the kernels' own time: ``trw_profile_prepare_last`` of the C ABI)."""
    return torch_rw_native.prepare_timing()


def set_rng_mode(mode):
    """Process-wide RNG mode used by ``walk`` (RNG_AUTO / RNG_GLIBC / RNG_WALKER)."""
    torch_rw_native.set_rng_mode(mode)


def get_rng_mode():
    return torch_rw_native.get_rng_mode()


def set_check(mode):
    """How the GPU calls report device-side data faults (degree-0 node, ids out of range, ...):

    ``"deferred"`` (default)  no host synchronisation, like the reference's launch-and-return GPU entry
                              point (csrc/cuda/rw_cuda.cu:223-245): the status block is read back
                              asynchronously and the ``RuntimeError`` is raised by the next torch_rw call
                              that finds it landed, or by ``check_status()``;
    ``True``                  strict: every call synchronises its stream and raises at once;
    ``False``                 never look (the output rows of a faulty walker are still defined).
    CPU tensors always raise at once (the host leg is synchronous anyway).
    """
    torch_rw_native.set_check(CHECK_DEFERRED if mode is None else _check_code(mode))


def get_check():
    return torch_rw_native.get_check()


def check_status(wait=True):
    """Raise ``RuntimeError`` for device-side faults of earlier deferred-check calls.  ``wait=True``
    first waits for every outstanding read-back; returns the number still in flight otherwise."""
    return torch_rw_native.check_status(wait)


def _report_pending_at_exit():
    """A deferred check nobody collected: the fault of a script's LAST torch_rw call (or of one in front of a long
    stretch without torch_rw calls) would otherwise never surface.  Non-blocking: only read-backs that
    have landed are looked at; the interpreter is going down, so a warning instead of an exception."""
    try:
        torch_rw_native.check_status(False)
    except RuntimeError as e:
        import warnings
        warnings.warn("torch_rw: a GPU call made under the deferred status check reported a data fault that no "
                      "later call raised: %s" % e, RuntimeWarning)
    except Exception:  # noqa: BLE001 -- never stand in the way of interpreter shutdown
        pass


import atexit as _atexit  # noqa: E402

_atexit.register(_report_pending_at_exit)


def walk_triples(triples_indexed, relation_tail_index, target_nodes, walk_length, padding_idx, seed, restart=True):
    """Knowledge-graph walks that emit (relation, tail) per hop: ``[N, 2*walk_length+1]`` int64.

    Same surface as the reference (torch_rw/rw.py:18-26, which reorders the arguments for the native
    call exactly like this): ``triples_indexed`` is ``[T, 3]`` (head, relation, tail) sorted by head and
    ``relation_tail_index`` ``[num_entities, 2]`` from ``utils.build_relation_tail_index``.  ``restart`` is
    accepted and unused, as in the reference (csrc/cpu/rw_cpu_triples.cpp:48-55).
    """
    return torch_rw_native.walk_triples(triples_indexed, relation_tail_index, target_nodes, walk_length,
                                        padding_idx, restart, seed)


def walk_triples_ex(triples_indexed, relation_tail_index, target_nodes, walk_length, padding_idx, seed,
                    restart=True, walker_base=0, rng_mode=RNG_AUTO, check=None):
    """``walk_triples`` with an explicit global walker base and RNG mode (see ``walk_ex``)."""
    return torch_rw_native.walk_triples_ex(triples_indexed, relation_tail_index, target_nodes, walk_length,
                                           padding_idx, restart, seed, walker_base, rng_mode, _check_code(check))


def walk_edge_list(edge_list_indexed, node_edge_index, target_nodes, p, q, walk_length, seed, padding_idx,
                   restart=True):
    """Walks over an edge list ``[E, 2]`` sorted by head + ``node_edge_index`` from
    ``utils.build_node_edge_index``; dead ends step to ``padding_idx`` and, if ``restart``, the walk
    then jumps back to its start node.  Same surface as the reference (torch_rw/rw.py:6-16)."""
    return torch_rw_native.walk_edge_list(edge_list_indexed, node_edge_index, target_nodes, p, q, walk_length, seed,
                                          padding_idx, restart)


def walk_edge_list_ex(edge_list_indexed, node_edge_index, target_nodes, p, q, walk_length, seed, padding_idx,
                      restart=True, walker_base=0, rng_mode=RNG_AUTO, check=None):
    """``walk_edge_list`` with an explicit global walker base and RNG mode (see ``walk_ex``)."""
    return torch_rw_native.walk_edge_list_ex(edge_list_indexed, node_edge_index, target_nodes, p, q, walk_length, seed,
                                             padding_idx, restart, walker_base, rng_mode, _check_code(check))


def to_windows(walks, window_size, num_nodes, seed):
    """Skip-gram windows over ``walks``: ``(target_nodes [W], pos_windows [W, ws-1], neg_windows [W, ws-1])``
    with ``W = num_walks * (walk_width - window_size + 1)``; negatives are ``rand() % num_nodes``.
    Same surface as the reference (torch_rw/rw.py:29-30); reference-exact negatives at any size."""
    return torch_rw_native.to_windows(walks, window_size, num_nodes, seed)


def walk_windows(row_ptr, col_idx, target_nodes, p, q, walk_length, seed, window_size, num_nodes, window_seed=None,
                 walker_base=0, return_walks=False, check=None):
    """``to_windows(walk(...), window_size, num_nodes, seed)`` as one call (torch_rw/rw.py:3-4 then :29-30).

    Returns ``(target_nodes, pos_windows, neg_windows)`` -- plus the ``[N, walk_length+1]`` walks as a fourth item
    with ``return_walks=True`` -- bit for bit what the two calls give (``window_seed``: the seed of the negatives,
    default ``seed`` as in the usual two-call use; ``walker_base``: global index of ``target_nodes[0]``, numbers both
    the walkers and their windows for a shard).  For the reference-exact uniform walk (``p == q == 1``) on the GPU the
    walk kernel writes the windows straight from its on-chip tile: the walks never travel to HBM and back (no
    reference counterpart; csrc/cpu/windows_cpu.cpp:5-77 consumes the [N, L+1] tensor).  node2vec walks, CPU
    tensors and ``window_size > 9`` run as the two calls."""
    out = torch_rw_native.walk_windows(row_ptr, col_idx, target_nodes, p, q, walk_length, seed, window_size, num_nodes,
                                       seed if window_seed is None else window_seed, walker_base, return_walks,
                                       _check_code(check))
    return out if return_walks else out[:3]


def to_windows_cbow(walks, window_size, num_nodes, seed):
    """CBOW windows: ``(pos_nodes [W], neg_nodes [W], windows [W, ws-1])``; the negative is redrawn while it
    equals the positive node.  Same surface as the reference (torch_rw/rw.py:32-33)."""
    return torch_rw_native.to_windows_cbow(walks, window_size, num_nodes, seed)


def to_windows_ex(walks, window_size, num_nodes, seed, walk_base=0):
    """``to_windows`` for a shard: ``walk_base`` = global index of ``walks[0]``."""
    return torch_rw_native.to_windows_ex(walks, window_size, num_nodes, seed, walk_base)


def to_windows_cbow_ex(walks, window_size, num_nodes, seed, window_base=0, rng_mode=RNG_AUTO):
    """``to_windows_cbow`` with an explicit global window base and RNG mode."""
    return torch_rw_native.to_windows_cbow_ex(walks, window_size, num_nodes, seed, window_base, rng_mode)


def to_windows_triples(walks, window_size, num_nodes, padding_idx, triples, seed):
    """Windows over triples walks: ``(target_triples [W,3], pos_windows [W,2ws,3], neg_windows [W,2ws,3])``,
    ``W = num_walks * ((walk_width-1)//2)``; negatives are uniformly drawn rows of ``triples``.
    Same surface as the reference (torch_rw/rw.py:35-36; ``num_nodes`` is unused there too)."""
    return torch_rw_native.to_windows_triples(walks, window_size, num_nodes, padding_idx, triples, seed)


def to_windows_triples_cbow(walks, window_size, num_nodes, padding_idx, triples, seed):
    """CBOW form: ``(pos_triples [W,3], neg_triples [W,3], pos_windows [W,2ws,3])``
    (torch_rw/rw.py:38-39)."""
    return torch_rw_native.to_windows_triples_cbow(walks, window_size, num_nodes, padding_idx, triples, seed)


def to_windows_triples_ex(walks, window_size, padding_idx, triples, seed, cbow=False, base=0, rng_mode=RNG_AUTO):
    """Shard/RNG-mode form of the two triples window builders (``base`` = global index of the first walk,
    or of the first window for the CBOW form)."""
    return torch_rw_native.to_windows_triples_ex(walks, window_size, padding_idx, triples, seed, cbow, base, rng_mode)
