// torch_rw_native -- the PyTorch-ROCm face of libtrw_hip.so.
//
// Plays the role of the reference's pybind dispatcher (csrc/rw_init.cpp:11-25 and :47-75,
// m.def("walk") / m.def("walk_triples") at :134/:136) for the walk paths built so far: same module name, same
// function name, same seven positional arguments.  Everything below the
// argument checks is the C ABI of include/trw.h; torch is used for device
// memory (output + scratch through the caching allocator), the device guard and
// the current stream.
//
// Device dispatch as in the reference (rw_init.cpp:20-24): tensors on the GPU go to the HIP kernels,
// CPU tensors of the CSR walk to the library's host leg (trw_walk_csr_cpu).  Neither is a fallback for
// the other: a GPU tensor never runs on the host.
//
// Data faults found on the device (degree-0 node, ids out of range, ...) are reported without a host
// synchronisation by default, like any asynchronous GPU error: the status block of a call is read back
// asynchronously and the error is raised by the NEXT torch_rw call (or by check_status()) once that
// read-back has landed.  set_check(True) / check=True is the strict mode: synchronise and raise at once.
#include <torch/extension.h>

// PyTorch-ROCm presents its HIP devices under the "cuda" device type; these are the guard and
// stream accessors that go with that.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <ATen/Parallel.h>

#include <atomic>
#include <string>

#include "../../include/trw.h"

namespace {

std::atomic<int> g_rng_mode{TRW_RNG_AUTO};
enum : int { kCheckOff = 0, kCheckStrict = 1, kCheckDeferred = 2 };
std::atomic<int> g_check{kCheckDeferred};

void check_index_tensor(const torch::Tensor &t, const char *name, bool want_cuda = true)
{
    // reference: CHECK_CUDA / CHECK_CPU (csrc/cuda/utils.cuh:7-8, used at rw_cuda.cu:194-196, rw_cpu.cpp:211-213)
    if (want_cuda) {
        TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor (the other tensors of this call are on the GPU)");
    } else {
        TORCH_CHECK(!t.is_cuda(), name, " must be a CPU tensor (the other tensors of this call are on the host)");
    }
    // reference: the int64 accessor raises this text for any other dtype
    TORCH_CHECK(t.scalar_type() == torch::kInt64, "expected scalar type Long but found ",
                c10::toString(t.scalar_type()), " (", name, ")");
    TORCH_CHECK(t.dim() == 1, name, " must be 1-dimensional, got ", t.dim(), " dimensions");
}

std::string status_text(int64_t bits)
{
    std::string s;
    if (bits & TRW_ST_DEGREE0) s += "a walker reached a node with no out-edges (the reference divides by zero there); ";
    if (bits & TRW_ST_NODE_RANGE) s += "a start node or col_idx entry is outside [0, num_nodes); ";
    if (bits & TRW_ST_BAD_CSR) s += "row_ptr is not a monotone sequence within [0, len(col_idx)]; ";
    if (bits & TRW_ST_TRIAL_CAP) s += "a rejection loop exceeded 2^20 trials at one step; ";
    return s;
}

// Deferred mode, on entry: raise what earlier calls' read-backs have reported by now (never waits).
void raise_pending()
{
    int64_t bits = 0;
    const int rc = trw_status_collect(0, &bits);
    TORCH_CHECK(rc >= 0, "torch_rw status collect failed (", rc, "): ", trw_last_error());
    TORCH_CHECK(bits == 0, "an earlier torch_rw call reported: ", status_text(bits), "status bits ", bits);
}

// After the kernels of a call are enqueued: look at its status block the way `mode` asks.
void finish_status(const torch::Tensor &workspace, void *stream, int mode, const char *what)
{
    if (mode == kCheckStrict) {
        int64_t status[4] = {0, 0, 0, 0};
        int rc;
        {
            pybind11::gil_scoped_release nogil;
            rc = trw_walk_status(workspace.data_ptr(), stream, status);
        }
        TORCH_CHECK(rc == TRW_OK, "torch_rw status read failed (", rc, "): ", trw_last_error());
        TORCH_CHECK(status[0] == 0, "torch_rw ", what, ": ", status_text(status[0]), "status bits ", status[0]);
    } else if (mode == kCheckDeferred) {
        const int rc = trw_status_post(workspace.data_ptr(), stream);
        TORCH_CHECK(rc == TRW_OK, "torch_rw status post failed (", rc, "): ", trw_last_error());
    }
}

int resolve_check(int64_t check) { return check < 0 ? g_check.load() : (int)check; }

// The host leg: what the reference's dispatcher does with CPU tensors (rw_init.cpp:22-23 -> walk_cpu).
torch::Tensor walk_cpu_leg(const torch::Tensor &row_ptr, const torch::Tensor &col_idx,
                           const torch::Tensor &target_nodes, double p, double q, int64_t walk_length, int64_t seed,
                           int64_t walker_base, int64_t rng_mode, const c10::optional<torch::Tensor> &out)
{
    check_index_tensor(row_ptr, "row_ptr", false);
    check_index_tensor(col_idx, "col_idx", false);
    check_index_tensor(target_nodes, "target_nodes", false);
    TORCH_CHECK(row_ptr.numel() >= 1, "row_ptr must have num_nodes + 1 entries");
    const auto rp = row_ptr.contiguous();
    const auto ci = col_idx.contiguous();
    const auto tg = target_nodes.contiguous();
    const int64_t n_walkers = tg.numel();
    torch::Tensor walks;
    if (out.has_value() && out->defined()) {
        walks = *out;
        TORCH_CHECK(walks.scalar_type() == torch::kInt64 && !walks.is_cuda() && walks.is_contiguous() &&
                        walks.dim() == 2 && walks.size(0) == n_walkers && walks.size(1) == walk_length + 1,
                    "out must be a contiguous int64 [len(target_nodes), walk_length + 1] tensor on the inputs' device");
    } else {
        walks = torch::empty({n_walkers, walk_length + 1}, rp.options());  // rw_cpu.cpp:216-217
    }
    const int threads = at::get_num_threads();
    int rc;
    {
        pybind11::gil_scoped_release nogil;  // the reference holds the GIL here; nothing below needs it
        rc = trw_walk_csr_cpu(rp.data_ptr<int64_t>(), rp.numel() - 1, ci.data_ptr<int64_t>(), ci.numel(),
                              tg.data_ptr<int64_t>(), n_walkers, walker_base, p, q, (int32_t)walk_length,
                              (int32_t)seed, (int32_t)rng_mode, walks.data_ptr<int64_t>(), threads);
    }
    TORCH_CHECK(rc == TRW_OK, "torch_rw walk failed (", rc, "): ", trw_last_error());
    return walks;
}

// walk_ex: the reference's seven arguments plus what this library adds (global walker base, RNG mode,
// status check, and an optional destination: a contiguous int64 [N, L+1] tensor to walk into -- e.g. a
// shard's slice of the gathered result).
torch::Tensor walk_ex(const torch::Tensor &row_ptr, const torch::Tensor &col_idx, const torch::Tensor &target_nodes,
                      double p, double q, int64_t walk_length, int64_t seed, int64_t walker_base, int64_t rng_mode,
                      int64_t check, const c10::optional<torch::Tensor> &out)
{
    TORCH_CHECK(walk_length >= 1, "walk_length must be >= 1, got ", walk_length);
    TORCH_CHECK(walk_length < (int64_t(1) << 31) && seed >= INT32_MIN && seed <= INT32_MAX,
                "walk_length/seed must fit the reference's int arguments");
    if (!row_ptr.is_cuda())  // rw_init.cpp:20: the device of row_ptr decides
        return walk_cpu_leg(row_ptr, col_idx, target_nodes, p, q, walk_length, seed, walker_base, rng_mode, out);
    const int check_mode = resolve_check(check);
    if (check_mode == kCheckDeferred) raise_pending();
    check_index_tensor(row_ptr, "row_ptr");
    check_index_tensor(col_idx, "col_idx");
    check_index_tensor(target_nodes, "target_nodes");
    TORCH_CHECK(row_ptr.device() == col_idx.device() && row_ptr.device() == target_nodes.device(),
                "row_ptr, col_idx and target_nodes must be on the same device");
    TORCH_CHECK(row_ptr.numel() >= 1, "row_ptr must have num_nodes + 1 entries");

    const c10::hip::HIPGuardMasqueradingAsCUDA guard(row_ptr.device());
    const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(row_ptr.device().index());

    // the reference's accessors honour strides; the C ABI wants dense arrays
    const auto rp = row_ptr.contiguous();
    const auto ci = col_idx.contiguous();
    const auto tg = target_nodes.contiguous();
    const int64_t n_nodes = rp.numel() - 1;
    const int64_t nnz = ci.numel();
    const int64_t n_walkers = tg.numel();

    // rw_cuda.cu:201-203: a fresh [N, L+1] int64 tensor on the inputs' device, unless the caller brought one
    torch::Tensor walks;
    if (out.has_value() && out->defined()) {
        walks = *out;
        TORCH_CHECK(walks.scalar_type() == torch::kInt64 && walks.device() == rp.device() && walks.is_contiguous() &&
                        walks.dim() == 2 && walks.size(0) == n_walkers && walks.size(1) == walk_length + 1,
                    "out must be a contiguous int64 [len(target_nodes), walk_length + 1] tensor on the inputs' device");
    } else {
        walks = torch::empty({n_walkers, walk_length + 1}, rp.options());
    }

    const size_t ws_bytes =
        trw_walk_csr_workspace_bytes(n_nodes, nnz, n_walkers, (int32_t)walk_length, p, q, (int32_t)rng_mode);
    auto workspace = torch::empty({(int64_t)ws_bytes}, rp.options().dtype(torch::kUInt8));

    int rc;
    {
        pybind11::gil_scoped_release nogil;
        rc = trw_walk_csr_hip(rp.data_ptr<int64_t>(), n_nodes, ci.data_ptr<int64_t>(), nnz, tg.data_ptr<int64_t>(),
                              n_walkers, walker_base, p, q, (int32_t)walk_length, (int32_t)seed, (int32_t)rng_mode,
                              walks.data_ptr<int64_t>(), workspace.data_ptr(), ws_bytes, (void *)stream.stream());
    }
    TORCH_CHECK(rc == TRW_OK, "torch_rw walk failed (", rc, "): ", trw_last_error());
    finish_status(workspace, (void *)stream.stream(), check_mode, "walk");
    return walks;
}

// The reference's signature, verbatim (rw_init.cpp:11-17).
torch::Tensor walk(const torch::Tensor &row_ptr, const torch::Tensor &column_idx, const torch::Tensor &target_nodes,
                   const double p, const double q, const int walk_length, const int seed)
{
    return walk_ex(row_ptr, column_idx, target_nodes, p, q, walk_length, seed, 0, g_rng_mode.load(), -1, c10::nullopt);
}

// walk_triples: reference dispatcher csrc/rw_init.cpp:47-75, same argument order.
torch::Tensor walk_triples_ex(const torch::Tensor &triples_indexed, const torch::Tensor &relation_tail_index,
                              const torch::Tensor &target_nodes, int64_t walk_length, int64_t padding_idx,
                              bool restart, int64_t seed, int64_t walker_base, int64_t rng_mode, int64_t check)
{
    (void)restart;  // accepted and never read by the reference either (rw_cpu_triples.cpp:48-55)
    const int check_mode = resolve_check(check);
    if (check_mode == kCheckDeferred) raise_pending();
    check_index_tensor(target_nodes, "target_nodes");
    for (const auto *t : {&triples_indexed, &relation_tail_index}) {
        TORCH_CHECK(t->is_cuda(), "triples_indexed and relation_tail_index must be CUDA tensors (torch_rw_native here "
                                  "is the MI355X HIP build; it has no CPU walker)");
        TORCH_CHECK(t->scalar_type() == torch::kInt64, "expected scalar type Long but found ",
                    c10::toString(t->scalar_type()));
        TORCH_CHECK(t->dim() == 2, "triples_indexed / relation_tail_index must be 2-dimensional");
    }
    TORCH_CHECK(triples_indexed.size(1) == 3, "triples_indexed must be [T, 3] (head, relation, tail)");
    TORCH_CHECK(relation_tail_index.size(1) == 2, "relation_tail_index must be [num_entities, 2]");
    TORCH_CHECK(triples_indexed.device() == target_nodes.device() && relation_tail_index.device() == target_nodes.device(),
                "triples_indexed, relation_tail_index and target_nodes must be on the same device");
    TORCH_CHECK(walk_length >= 1, "walk_length must be >= 1, got ", walk_length);
    TORCH_CHECK(walk_length < (int64_t(1) << 30) && seed >= INT32_MIN && seed <= INT32_MAX,
                "walk_length/seed must fit the reference's int arguments");

    const c10::hip::HIPGuardMasqueradingAsCUDA guard(target_nodes.device());
    const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(target_nodes.device().index());
    const auto tr = triples_indexed.contiguous();
    const auto ix = relation_tail_index.contiguous();
    const auto tg = target_nodes.contiguous();
    const int64_t n_triples = tr.size(0), n_entities = ix.size(0), n_walkers = tg.numel();

    // rw_cuda_triples.cu: a fresh [N, 2L+1] int64 tensor on the inputs' device
    auto walks = torch::empty({n_walkers, 2 * walk_length + 1}, tg.options());
    const size_t ws_bytes = trw_walk_triples_workspace_bytes(n_triples, n_entities, n_walkers, (int32_t)walk_length,
                                                             padding_idx, (int32_t)rng_mode);
    auto workspace = torch::empty({(int64_t)ws_bytes}, tg.options().dtype(torch::kUInt8));
    int rc;
    {
        pybind11::gil_scoped_release nogil;
        rc = trw_walk_triples_hip(tr.data_ptr<int64_t>(), n_triples, ix.data_ptr<int64_t>(), n_entities,
                                  tg.data_ptr<int64_t>(), n_walkers, walker_base, (int32_t)walk_length, padding_idx,
                                  (int32_t)seed, (int32_t)rng_mode, walks.data_ptr<int64_t>(), workspace.data_ptr(),
                                  ws_bytes, (void *)stream.stream());
    }
    TORCH_CHECK(rc == TRW_OK, "torch_rw walk_triples failed (", rc, "): ", trw_last_error());
    finish_status(workspace, (void *)stream.stream(), check_mode, "walk_triples");
    return walks;
}

torch::Tensor walk_triples(const torch::Tensor &triples_indexed, const torch::Tensor &relation_tail_index,
                           const torch::Tensor &target_nodes, const int walk_length, const int64_t padding_idx,
                           const bool restart, const int seed)
{
    return walk_triples_ex(triples_indexed, relation_tail_index, target_nodes, walk_length, padding_idx, restart, seed,
                           0, g_rng_mode.load(), -1);
}

// walk_edge_list: reference dispatcher csrc/rw_init.cpp:27-45, same argument order.
torch::Tensor walk_edge_list_ex(const torch::Tensor &edge_list_indexed, const torch::Tensor &node_edges_idx,
                                const torch::Tensor &target_nodes, double p, double q, int64_t walk_length,
                                int64_t seed, int64_t padding_idx, bool restart, int64_t walker_base,
                                int64_t rng_mode, int64_t check)
{
    const int check_mode = resolve_check(check);
    if (check_mode == kCheckDeferred) raise_pending();
    check_index_tensor(target_nodes, "target_nodes");
    for (const auto *t : {&edge_list_indexed, &node_edges_idx}) {
        TORCH_CHECK(t->is_cuda(), "edge_list_indexed and node_edge_index must be CUDA tensors (torch_rw_native here "
                                  "is the MI355X HIP build; it has no CPU walker)");
        TORCH_CHECK(t->scalar_type() == torch::kInt64, "expected scalar type Long but found ",
                    c10::toString(t->scalar_type()));
        TORCH_CHECK(t->dim() == 2 && t->size(1) == 2, "edge_list_indexed / node_edge_index must be [*, 2]");
    }
    TORCH_CHECK(edge_list_indexed.device() == target_nodes.device() && node_edges_idx.device() == target_nodes.device(),
                "edge_list_indexed, node_edge_index and target_nodes must be on the same device");
    TORCH_CHECK(walk_length >= 1, "walk_length must be >= 1, got ", walk_length);
    TORCH_CHECK(walk_length < (int64_t(1) << 31) && seed >= INT32_MIN && seed <= INT32_MAX,
                "walk_length/seed must fit the reference's int arguments");

    const c10::hip::HIPGuardMasqueradingAsCUDA guard(target_nodes.device());
    const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(target_nodes.device().index());
    const auto ed = edge_list_indexed.contiguous();
    const auto ix = node_edges_idx.contiguous();
    const auto tg = target_nodes.contiguous();
    const int64_t n_edges = ed.size(0), n_nodes = ix.size(0), n_walkers = tg.numel();
    auto walks = torch::empty({n_walkers, walk_length + 1}, tg.options());
    const size_t ws_bytes = trw_walk_edge_list_workspace_bytes(n_edges, n_nodes, n_walkers, (int32_t)walk_length,
                                                               padding_idx, (int32_t)rng_mode);
    auto workspace = torch::empty({(int64_t)ws_bytes}, tg.options().dtype(torch::kUInt8));
    int rc;
    {
        pybind11::gil_scoped_release nogil;
        rc = trw_walk_edge_list_hip(ed.data_ptr<int64_t>(), n_edges, ix.data_ptr<int64_t>(), n_nodes,
                                    tg.data_ptr<int64_t>(), n_walkers, walker_base, p, q, (int32_t)walk_length,
                                    (int32_t)seed, padding_idx, restart ? 1 : 0, (int32_t)rng_mode,
                                    walks.data_ptr<int64_t>(), workspace.data_ptr(), ws_bytes,
                                    (void *)stream.stream());
    }
    TORCH_CHECK(rc == TRW_OK, "torch_rw walk_edge_list failed (", rc, "): ", trw_last_error());
    finish_status(workspace, (void *)stream.stream(), check_mode, "walk_edge_list");
    return walks;
}

torch::Tensor walk_edge_list(const torch::Tensor &edge_list_indexed, const torch::Tensor &node_edges_idx,
                             const torch::Tensor &target_nodes, const double p, const double q, const int walk_length,
                             const int seed, const int64_t padding_idx, const bool restart)
{
    return walk_edge_list_ex(edge_list_indexed, node_edges_idx, target_nodes, p, q, walk_length, seed, padding_idx,
                             restart, 0, g_rng_mode.load(), -1);
}

// to_windows / to_windows_cbow: reference dispatchers csrc/rw_init.cpp:77-101.
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> windows_impl(const torch::Tensor &walks, int64_t window_size,
                                                                     int64_t num_nodes, int64_t seed, bool cbow,
                                                                     int64_t base, int64_t rng_mode)
{
    TORCH_CHECK(walks.is_cuda(), "walks must be a CUDA tensor (torch_rw_native here is the MI355X HIP build; it has "
                                 "no CPU path)");
    TORCH_CHECK(walks.scalar_type() == torch::kInt64, "expected scalar type Long but found ",
                c10::toString(walks.scalar_type()));
    TORCH_CHECK(walks.dim() == 2, "walks must be 2-dimensional");
    TORCH_CHECK(walks.is_contiguous(), "walks must be a contigous tensor");  // windows_cpu.cpp:15 (sic)
    const int64_t num_walks = walks.size(0), cols = walks.size(1);
    TORCH_CHECK(window_size >= 1 && window_size <= cols, "window_size must be between 1 and the walk width");
    TORCH_CHECK(seed >= INT32_MIN && seed <= INT32_MAX, "seed must fit the reference's int argument");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(walks.device());
    const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(walks.device().index());
    const int64_t W = (cols - window_size + 1) * num_walks;
    auto centre = torch::empty({W}, walks.options());
    auto rest = torch::empty({W, window_size - 1}, walks.options());
    auto neg = cbow ? torch::empty({W}, walks.options()) : torch::empty({W, window_size - 1}, walks.options());
    const size_t ws_bytes = trw_to_windows_workspace_bytes(num_walks, cols, (int32_t)window_size);
    auto workspace = torch::empty({(int64_t)ws_bytes}, walks.options().dtype(torch::kUInt8));
    int rc;
    {
        pybind11::gil_scoped_release nogil;
        if (cbow)
            rc = trw_to_windows_cbow_hip(walks.data_ptr<int64_t>(), num_walks, cols, (int32_t)window_size, num_nodes,
                                         (int32_t)seed, base, (int32_t)rng_mode, centre.data_ptr<int64_t>(),
                                         neg.data_ptr<int64_t>(), rest.data_ptr<int64_t>(), workspace.data_ptr(),
                                         ws_bytes, (void *)stream.stream());
        else
            rc = trw_to_windows_hip(walks.data_ptr<int64_t>(), num_walks, cols, (int32_t)window_size, num_nodes,
                                    (int32_t)seed, base, centre.data_ptr<int64_t>(), rest.data_ptr<int64_t>(),
                                    neg.data_ptr<int64_t>(), workspace.data_ptr(), ws_bytes, (void *)stream.stream());
    }
    TORCH_CHECK(rc == TRW_OK, "torch_rw to_windows failed (", rc, "): ", trw_last_error());
    if (cbow) return std::make_tuple(centre, neg, rest);   // (pos_nodes, neg_nodes, windows), windows_cpu.cpp:157
    return std::make_tuple(centre, rest, neg);             // (target_nodes, pos_windows, neg_windows), :76
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> to_windows(const torch::Tensor &walks, const int window_size,
                                                                   const int64_t num_nodes, const int seed)
{
    return windows_impl(walks, window_size, num_nodes, seed, false, 0, TRW_RNG_GLIBC);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> to_windows_cbow(const torch::Tensor &walks,
                                                                        const int window_size,
                                                                        const int64_t num_nodes, const int seed)
{
    return windows_impl(walks, window_size, num_nodes, seed, true, 0, g_rng_mode.load());
}

// to_windows_triples / to_windows_triples_cbow: reference dispatchers csrc/rw_init.cpp:103-131.
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> triples_windows_impl(const torch::Tensor &walks,
                                                                             int64_t window_size, int64_t padding_idx,
                                                                             const torch::Tensor &triples, int64_t seed,
                                                                             bool cbow, int64_t base, int64_t rng_mode)
{
    for (const auto *t : {&walks, &triples}) {
        TORCH_CHECK(t->is_cuda(), "walks and triples must be CUDA tensors (torch_rw_native here is the MI355X HIP "
                                  "build; it has no CPU path)");
        TORCH_CHECK(t->scalar_type() == torch::kInt64, "expected scalar type Long but found ",
                    c10::toString(t->scalar_type()));
        TORCH_CHECK(t->dim() == 2, "walks and triples must be 2-dimensional");
    }
    TORCH_CHECK(walks.is_contiguous(), "walks must be a contigous tensor");  // windows_cpu.cpp:172 (sic)
    TORCH_CHECK(triples.size(1) == 3 && triples.size(0) >= 1, "triples must be [T >= 1, 3]");
    TORCH_CHECK(walks.device() == triples.device(), "walks and triples must be on the same device");
    TORCH_CHECK(window_size >= 1, "window_size must be >= 1");
    TORCH_CHECK(seed >= INT32_MIN && seed <= INT32_MAX, "seed must fit the reference's int argument");
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(walks.device());
    const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(walks.device().index());
    const auto tr = triples.contiguous();
    const int64_t num_walks = walks.size(0), cols = walks.size(1);
    const int64_t W = ((cols - 1) / 2) * num_walks;
    auto centre = torch::empty({W, 3}, walks.options());
    auto pos = torch::empty({W, 2 * window_size, 3}, walks.options());
    auto neg = cbow ? torch::empty({W, 3}, walks.options()) : torch::empty({W, 2 * window_size, 3}, walks.options());
    const size_t ws_bytes = trw_to_windows_triples_workspace_bytes(num_walks, cols, (int32_t)window_size);
    auto workspace = torch::empty({(int64_t)ws_bytes}, walks.options().dtype(torch::kUInt8));
    int rc;
    {
        pybind11::gil_scoped_release nogil;
        if (cbow)
            rc = trw_to_windows_triples_cbow_hip(walks.data_ptr<int64_t>(), num_walks, cols, (int32_t)window_size,
                                                 padding_idx, tr.data_ptr<int64_t>(), tr.size(0), (int32_t)seed, base,
                                                 (int32_t)rng_mode, centre.data_ptr<int64_t>(),
                                                 neg.data_ptr<int64_t>(), pos.data_ptr<int64_t>(),
                                                 workspace.data_ptr(), ws_bytes, (void *)stream.stream());
        else
            rc = trw_to_windows_triples_hip(walks.data_ptr<int64_t>(), num_walks, cols, (int32_t)window_size,
                                            padding_idx, tr.data_ptr<int64_t>(), tr.size(0), (int32_t)seed, base,
                                            centre.data_ptr<int64_t>(), pos.data_ptr<int64_t>(),
                                            neg.data_ptr<int64_t>(), workspace.data_ptr(), ws_bytes,
                                            (void *)stream.stream());
    }
    TORCH_CHECK(rc == TRW_OK, "torch_rw to_windows_triples failed (", rc, "): ", trw_last_error());
    if (cbow) return std::make_tuple(centre, neg, pos);   // (pos_triples, neg_triples, pos_windows), windows_cpu.cpp:473
    return std::make_tuple(centre, pos, neg);             // (target_triples, pos_windows, neg_windows), :313
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> to_windows_triples(const torch::Tensor &walks,
                                                                           const int window_size,
                                                                           const int64_t num_nodes,
                                                                           const int64_t padding_idx,
                                                                           const torch::Tensor &triples, const int seed)
{
    (void)num_nodes;  // unused by the reference as well
    return triples_windows_impl(walks, window_size, padding_idx, triples, seed, false, 0, TRW_RNG_GLIBC);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> to_windows_triples_cbow(const torch::Tensor &walks,
                                                                                const int window_size,
                                                                                const int64_t num_nodes,
                                                                                const int64_t padding_idx,
                                                                                const torch::Tensor &triples,
                                                                                const int seed)
{
    (void)num_nodes;
    return triples_windows_impl(walks, window_size, padding_idx, triples, seed, true, 0, g_rng_mode.load());
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "MI355X-native CSR random walks behind the torch_rw.rw.walk surface";
    m.def("walk", &walk, "walk", pybind11::arg("row_ptr"), pybind11::arg("col_idx"), pybind11::arg("target_nodes"),
          pybind11::arg("p"), pybind11::arg("q"), pybind11::arg("walk_length"), pybind11::arg("seed"));
    m.def("walk_ex", &walk_ex, "walk with an explicit global walker base, RNG mode and status check",
          pybind11::arg("row_ptr"), pybind11::arg("col_idx"), pybind11::arg("target_nodes"), pybind11::arg("p"),
          pybind11::arg("q"), pybind11::arg("walk_length"), pybind11::arg("seed"), pybind11::arg("walker_base") = 0,
          pybind11::arg("rng_mode") = (int64_t)TRW_RNG_AUTO, pybind11::arg("check") = (int64_t)-1,
          pybind11::arg("out") = pybind11::none());
    m.def("to_windows", &to_windows, "to_windows", pybind11::arg("walks"), pybind11::arg("window_size"),
          pybind11::arg("num_nodes"), pybind11::arg("seed"));
    m.def("to_windows_cbow", &to_windows_cbow, "to_windows", pybind11::arg("walks"), pybind11::arg("window_size"),
          pybind11::arg("num_nodes"), pybind11::arg("seed"));
    m.def("to_windows_triples", &to_windows_triples, "to_windows_triples", pybind11::arg("walks"),
          pybind11::arg("window_size"), pybind11::arg("num_nodes"), pybind11::arg("padding_idx"),
          pybind11::arg("triples"), pybind11::arg("seed"));
    m.def("to_windows_triples_cbow", &to_windows_triples_cbow, "to_windows_triples_cbow", pybind11::arg("walks"),
          pybind11::arg("window_size"), pybind11::arg("num_nodes"), pybind11::arg("padding_idx"),
          pybind11::arg("triples"), pybind11::arg("seed"));
    m.def("to_windows_triples_ex", [](const torch::Tensor &walks, int64_t window_size, int64_t padding_idx,
                                      const torch::Tensor &triples, int64_t seed, bool cbow, int64_t base,
                                      int64_t rng_mode) {
        return triples_windows_impl(walks, window_size, padding_idx, triples, seed, cbow, base,
                                    cbow ? rng_mode : (int64_t)TRW_RNG_GLIBC);
    }, pybind11::arg("walks"), pybind11::arg("window_size"), pybind11::arg("padding_idx"), pybind11::arg("triples"),
          pybind11::arg("seed"), pybind11::arg("cbow") = false, pybind11::arg("base") = 0,
          pybind11::arg("rng_mode") = (int64_t)TRW_RNG_AUTO);
    m.def("to_windows_ex", [](const torch::Tensor &walks, int64_t window_size, int64_t num_nodes, int64_t seed,
                              int64_t walk_base) {
        return windows_impl(walks, window_size, num_nodes, seed, false, walk_base, TRW_RNG_GLIBC);
    }, pybind11::arg("walks"), pybind11::arg("window_size"), pybind11::arg("num_nodes"), pybind11::arg("seed"),
          pybind11::arg("walk_base") = 0);
    m.def("to_windows_cbow_ex", [](const torch::Tensor &walks, int64_t window_size, int64_t num_nodes, int64_t seed,
                                   int64_t window_base, int64_t rng_mode) {
        return windows_impl(walks, window_size, num_nodes, seed, true, window_base, rng_mode);
    }, pybind11::arg("walks"), pybind11::arg("window_size"), pybind11::arg("num_nodes"), pybind11::arg("seed"),
          pybind11::arg("window_base") = 0, pybind11::arg("rng_mode") = (int64_t)TRW_RNG_AUTO);
    m.def("walk_edge_list", &walk_edge_list, "walk_edge_list", pybind11::arg("edge_list_indexed"),
          pybind11::arg("node_edge_index"), pybind11::arg("target_nodes"), pybind11::arg("p"), pybind11::arg("q"),
          pybind11::arg("walk_length"), pybind11::arg("seed"), pybind11::arg("padding_idx"), pybind11::arg("restart"));
    m.def("walk_edge_list_ex", &walk_edge_list_ex, "walk_edge_list with a global walker base, RNG mode, status check",
          pybind11::arg("edge_list_indexed"), pybind11::arg("node_edge_index"), pybind11::arg("target_nodes"),
          pybind11::arg("p"), pybind11::arg("q"), pybind11::arg("walk_length"), pybind11::arg("seed"),
          pybind11::arg("padding_idx"), pybind11::arg("restart"), pybind11::arg("walker_base") = 0,
          pybind11::arg("rng_mode") = (int64_t)TRW_RNG_AUTO, pybind11::arg("check") = (int64_t)-1);
    m.def("walk_triples", &walk_triples, "walk_triples", pybind11::arg("triples_indexed"),
          pybind11::arg("relation_tail_index"), pybind11::arg("target_nodes"), pybind11::arg("walk_length"),
          pybind11::arg("padding_idx"), pybind11::arg("restart"), pybind11::arg("seed"));
    m.def("walk_triples_ex", &walk_triples_ex, "walk_triples with a global walker base, RNG mode and status check",
          pybind11::arg("triples_indexed"), pybind11::arg("relation_tail_index"), pybind11::arg("target_nodes"),
          pybind11::arg("walk_length"), pybind11::arg("padding_idx"), pybind11::arg("restart"), pybind11::arg("seed"),
          pybind11::arg("walker_base") = 0, pybind11::arg("rng_mode") = (int64_t)TRW_RNG_AUTO,
          pybind11::arg("check") = (int64_t)-1);
    m.def("set_rng_mode", [](int64_t mode) {
        TORCH_CHECK(mode == TRW_RNG_AUTO || mode == TRW_RNG_GLIBC || mode == TRW_RNG_WALKER, "unknown rng mode ", mode);
        g_rng_mode.store((int)mode);
    });
    m.def("get_rng_mode", []() { return (int64_t)g_rng_mode.load(); });
    m.def("set_check", [](int64_t mode) {
        TORCH_CHECK(mode == kCheckOff || mode == kCheckStrict || mode == kCheckDeferred, "unknown check mode ", mode);
        g_check.store((int)mode);
    });
    m.def("get_check", []() { return (int64_t)g_check.load(); });
    m.def("check_status", [](bool wait) {
        int64_t bits = 0;
        int rc;
        {
            pybind11::gil_scoped_release nogil;
            rc = trw_status_collect(wait ? 1 : 0, &bits);
        }
        TORCH_CHECK(rc >= 0, "torch_rw status collect failed (", rc, "): ", trw_last_error());
        TORCH_CHECK(bits == 0, "torch_rw: ", status_text(bits), "status bits ", bits);
        return (int64_t)rc;
    }, pybind11::arg("wait") = true);
    m.attr("CHECK_OFF") = (int64_t)kCheckOff;
    m.attr("CHECK_STRICT") = (int64_t)kCheckStrict;
    m.attr("CHECK_DEFERRED") = (int64_t)kCheckDeferred;
    m.def("abi_version", []() { return (int64_t)trw_version(); });
    m.attr("RNG_AUTO") = (int64_t)TRW_RNG_AUTO;
    m.attr("RNG_GLIBC") = (int64_t)TRW_RNG_GLIBC;
    m.attr("RNG_WALKER") = (int64_t)TRW_RNG_WALKER;
}
