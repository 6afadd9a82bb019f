// torch_rw_native -- the PyTorch-ROCm face of libtrw_hip.so.
//
// Plays the role of the reference's pybind dispatcher (csrc/rw_init.cpp:11-25 and :47-75,
// m.def("walk") / m.def("walk_triples") at :134/:136) for the walk paths built so far: same module name, same
// function name, same seven positional arguments.  Everything below the
// argument checks is the C ABI of include/trw.h; torch is used for device
// memory (output + scratch through the caching allocator), the device guard and
// the current stream.
//
// Device dispatch as in the reference (rw_init.cpp:20-24 and the five other dispatchers): tensors on the GPU go to
// the HIP kernels, CPU tensors to the library's host legs (trw_*_cpu: csrc/cpu on one thread).  Neither is a fallback for
// the other: a GPU tensor never runs on the host.
//
// Data faults found on the device (degree-0 node, ids out of range, ...) are reported without a host
// synchronisation by default, like any asynchronous GPU error: the status block of a call is read back
// asynchronously and the error is raised by the NEXT torch_rw call (or by check_status()) once that
// read-back has landed.  set_check(True) / check=True is the strict mode: synchronise and raise at once.
#include <torch/extension.h>
#include <atomic>
#include <chrono>

// PyTorch-ROCm presents its HIP devices under the "cuda" device type; these are the guard and
// stream accessors that go with that.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <ATen/Parallel.h>

#include <atomic>
#include <cstdlib>
#include <list>
#include <mutex>
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
    torch::Tensor walks, narrow;  // (an int32 `out`: narrow rows -- walked as int64 rows here, narrowed on the way out)
    if (out.has_value() && out->defined()) {
        walks = *out;
        TORCH_CHECK((walks.scalar_type() == torch::kInt64 || walks.scalar_type() == torch::kInt32) && !walks.is_cuda() &&
                        walks.is_contiguous() && walks.dim() == 2 && walks.size(0) == n_walkers &&
                        walks.size(1) == walk_length + 1,
                    "out must be a contiguous int64 (or, for narrow rows, int32) [len(target_nodes), walk_length + 1] tensor "
                    "on the inputs' device");
        if (walks.scalar_type() == torch::kInt32) {
            TORCH_CHECK(rp.numel() - 1 <= (int64_t(1) << 32), "narrow rows hold node ids below 2^32");
            narrow = walks;
            walks = torch::empty({n_walkers, walk_length + 1}, rp.options());
        }
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
    if (narrow.defined()) {
        int64_t overflow = 0;
        rc = trw_rows_narrow_cpu(walks.data_ptr<int64_t>(), walks.numel(), (uint32_t *)narrow.data_ptr<int32_t>(), &overflow);
        TORCH_CHECK(rc == TRW_OK, "torch_rw narrow_rows failed (", rc, "): ", trw_last_error());
        return narrow;
    }
    return walks;
}

// Host-side cost of the last prepared-graph build (rw.prepare_timing): what the allocator took and what enqueueing the
// kernels took -- beside the kernels' own time (trw_profile_prepare_last) that is all a prepare call is.
static std::atomic<double> g_last_alloc_ms{0.0}, g_last_launch_ms{0.0};
static std::atomic<long long> g_last_keep_bytes{0}, g_last_scratch_bytes{0};

// ---- prepared graphs, kept across calls --------------------------------------------------------------------
// What a walk derives from (row_ptr, col_idx) alone (uint32 shadow, fat entries, edge table, neighbourhood
// masks: trw_prepare_csr_hip) is cached per graph, so the second and later walks over the same tensors start at
// the walk kernel (SURVEY 8(b) "Ownership": cached derived data keyed on data_ptr/version).  An entry is valid for
// exactly the storages it was built from, while they are alive (weak references: a freed-and-reused address can
// not match) and unmodified (tensor version counters: an in-place write invalidates).  Policy: the first walk
// builds what a single call of its shape is worth; later node2vec walks over the same graph also build the
// neighbourhood masks of longer rows (TRW_PREP_MARKS, TRW_PREP_MARKS_FULL) once the steps made over the graph
// warrant the pass (kMarksAfterStepsPerEntry below); rw.prepare() asks for them up front (marks=True / marks="full").  Least-recently-used graphs beyond `capacity` are dropped; entries whose
// tensors died are dropped at the next call.  TRW_CACHE=0 (or set_cache(False)) turns the cache off: every
// call then prepares into its own workspace, as trw_walk_csr_hip does.  Calls on a stream that is being captured
// into a HIP graph never use the cache (a replay must not depend on memory the cache may drop).
struct PreparedGraph {
    c10::weak_intrusive_ptr<c10::StorageImpl> rp_storage, ci_storage;
    const void *rp_ptr = nullptr, *ci_ptr = nullptr;
    int64_t rp_numel = 0, ci_numel = 0;
    uint32_t rp_version = 0, ci_version = 0;
    c10::Device device{c10::kCPU};
    int kind = 0;                 // 0: CSR (row_ptr, col_idx); 1: edge list (edge_list_indexed, node_edge_index)
    int64_t pad = 0;              // edge list: the padding id the shadow was built for
    int32_t what = 0;
    torch::Tensor blob;
    void *ready = nullptr;        // event recorded after the prepare kernels
    void *built_on = nullptr;     // the stream they ran on
    int64_t biased_walks = 0;
    int64_t biased_steps = 0;     // steps of node2vec walks (per-walker stream) made over this graph so far
    int64_t uniform_steps = 0;    // steps of reference-exact uniform walks made over this graph so far
    int32_t refused = 0;          // TRW_PREP_* bits whose allocation failed once: not asked for again
    bool pinned = false;          // rw.prepare(): not upgraded/evicted behind the caller's back
    PreparedGraph(c10::weak_intrusive_ptr<c10::StorageImpl> a, c10::weak_intrusive_ptr<c10::StorageImpl> b)
        : rp_storage(std::move(a)), ci_storage(std::move(b)) {}
    ~PreparedGraph() { trw_event_destroy(ready); }
};

// When a graph that is walked again gets more of its neighbourhood masks, in node2vec steps already made over it per
// CSR entry.  A first walk that is worth the entries brings the masks of rows up to 256 entries along (trw_prepare_plan);
// those of rows up to 1 024 take a REBUILD of the prepared graph here -- 0.42 ns per entry on R-MAT 10M/110M (round 5) for
// 0.011 ns off every later step: from 40 steps per entry on (the sixth batch of one 80-step walker per node of a graph of
// average degree 11), and not at all by a stream of 400-walker batches (a 46 ms pass in front of a 3 ms walk).
constexpr int64_t kMarksAfterStepsPerEntry = 6, kFullMarksAfterStepsPerEntry = 40;
std::mutex g_cache_mu;
std::list<std::shared_ptr<PreparedGraph>> g_cache;  // most recently used first
std::atomic<int64_t> g_cache_capacity{2};
std::atomic<int> g_cache_on{-1};  // -1: not decided yet (TRW_CACHE)

bool cache_enabled()
{
    int v = g_cache_on.load();
    if (v < 0) {
        const char *env = std::getenv("TRW_CACHE");
        v = (env && env[0] == '0') ? 0 : 1;
        g_cache_on.store(v);
    }
    return v == 1;
}

bool cacheable(const torch::Tensor &t) { return !t.is_inference() && t.has_storage(); }

// Same graph MEMORY (storages alive and identical, same window into them, same kind / padding id, same device) ...
bool entry_same_memory(const PreparedGraph &e, const torch::Tensor &rp, const torch::Tensor &ci, int kind, int64_t pad)
{
    const auto a = e.rp_storage.lock(), b = e.ci_storage.lock();
    return e.kind == kind && e.pad == pad && a && b && a.get() == rp.storage().unsafeGetStorageImpl() && b.get() == ci.storage().unsafeGetStorageImpl() &&
           e.rp_ptr == rp.data_ptr() && e.ci_ptr == ci.data_ptr() && e.rp_numel == rp.numel() &&
           e.ci_numel == ci.numel() && e.device == rp.device();
}

// ... and same CONTENT as far as torch can tell (no in-place write since the entry was built).
bool entry_matches(const PreparedGraph &e, const torch::Tensor &rp, const torch::Tensor &ci, int kind, int64_t pad)
{
    return entry_same_memory(e, rp, ci, kind, pad) && e.rp_version == rp._version() && e.ci_version == ci._version();
}

// The eviction rule, on its own so that it can be tested without a GPU (`_cache_victims`): the cache holds `capacity`
// graphs PER DEVICE (a process that drives several GPUs keeps each card's working set).  Entries
// are listed most recently used first, [0] being the current call's; among the entries of [0]'s device beyond the
// capacity the least recently used unpinned ones go, never [0] itself.
std::vector<int64_t> cache_victims(const std::vector<int64_t> &device_of, const std::vector<bool> &pinned, int64_t capacity)
{
    std::vector<int64_t> victims;
    if (device_of.empty()) return victims;
    const int64_t cap = capacity < 1 ? 1 : capacity, dev = device_of[0];
    int64_t on_device = 0;
    for (const int64_t d : device_of) on_device += d == dev ? 1 : 0;
    for (int64_t i = (int64_t)device_of.size() - 1; i >= 1 && on_device > cap; --i) {
        if (device_of[i] != dev || pinned[i]) continue;
        victims.push_back(i);
        --on_device;
    }
    return victims;
}

// (Re)builds the prepared blob of `entry` with `what`; falls back to leaner layouts when the device is short of
// memory (each part is optional: the walk then takes the kernel that does without it).
void build_blob(PreparedGraph &entry, const torch::Tensor &rp, const torch::Tensor &ci, int32_t what, void *stream)
{
    const bool elist = entry.kind == 1;  // rp = edge_list_indexed [E, 2], ci = node_edge_index [n, 2]
    const int64_t n_nodes = elist ? ci.size(0) : rp.numel() - 1, nnz = elist ? rp.size(0) : ci.numel();
    // The old blob goes first: an upgrade rebuilds everything, and on a graph the size of BASELINE configs[3] old and
    // new blob (131 GB each) do not fit side by side.  (A call still walking over the old one holds its own reference.)
    entry.blob = torch::Tensor();
    entry.what = -100;
    const int32_t csr_ladder[4] = {what, (int32_t)(what & ~(TRW_PREP_MARKS | TRW_PREP_MARKS_FULL)),
                                   (int32_t)(what & TRW_PREP_UNIFORM), 0};
    const int32_t el_ladder[4] = {what, (int32_t)(what & TRW_ELPREP_TABLE), 0, 0};
    const int32_t *ladder = elist ? el_ladder : csr_ladder;
    for (int k = 0; k < 4; ++k) {
        const int32_t w = ladder[k];
        if (k > 0 && w == ladder[k - 1]) continue;
        // What the walks read is kept; the marking passes' scratch (the in-lists' items, the tile pass's task records: 10 bytes
        // per entry) is a temporary of this call -- it goes back to the allocator when the call returns, in stream order.
        const size_t bytes = elist ? trw_prepare_edge_list_bytes(nnz, n_nodes, entry.pad, w)
                                   : trw_prepare_csr_persistent_bytes(n_nodes, nnz, w);
        const size_t scratch_bytes = elist ? 0 : trw_prepare_csr_scratch_bytes(n_nodes, nnz, w);
        torch::Tensor blob, scratch;
        const auto t_alloc = std::chrono::steady_clock::now();
        try {
            blob = torch::empty({(int64_t)bytes}, rp.options().dtype(torch::kUInt8));
            if (scratch_bytes) scratch = torch::empty({(int64_t)scratch_bytes}, rp.options().dtype(torch::kUInt8));
        } catch (const c10::OutOfMemoryError &) {
            if (k == 3) throw;
            continue;  // (the allocator has already given its cached blocks back before throwing)
        }
        const auto t_launch = std::chrono::steady_clock::now();
        const int rc = elist ? trw_prepare_edge_list_hip(rp.data_ptr<int64_t>(), nnz, ci.data_ptr<int64_t>(), n_nodes,
                                                         entry.pad, w, blob.data_ptr(), bytes, stream)
                             : trw_prepare_csr_scratch_hip(rp.data_ptr<int64_t>(), n_nodes, ci.data_ptr<int64_t>(), nnz, w,
                                                           blob.data_ptr(), bytes, scratch_bytes ? scratch.data_ptr() : nullptr,
                                                           scratch_bytes, stream);
        TORCH_CHECK(rc == TRW_OK, "torch_rw prepare failed (", rc, "): ", trw_last_error());
        g_last_alloc_ms = std::chrono::duration<double, std::milli>(t_launch - t_alloc).count();
        g_last_launch_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_launch).count();
        g_last_keep_bytes = (long long)bytes;
        g_last_scratch_bytes = (long long)scratch_bytes;
        trw_event_destroy(entry.ready);
        entry.ready = trw_event_record(stream);
        entry.built_on = stream;
        entry.blob = blob;
        entry.what = w;
        entry.refused |= what & ~w;  // what did not fit is not tried again for this graph
        return;
    }
}

// What one call takes from the cache: its own reference to the blob (an upgrade by another thread replaces the
// entry's tensor, never the memory under a running call) and the layout flags that go with it.
struct PreparedUse {
    torch::Tensor blob;
    int32_t what = 0;
    bool valid = false;
};

// The prepared graph for this call (built or upgraded as needed); !valid when the call should prepare for
// itself (cache off, capture in progress, uncacheable tensors).
PreparedUse prepared_for(const torch::Tensor &rp, const torch::Tensor &ci, int32_t needed, bool biased_walk, bool pin,
                         const c10::hip::HIPStreamMasqueradingAsCUDA &torch_stream, int kind = 0, int64_t pad = 0,
                         int64_t uniform_steps = 0, int64_t biased_steps = 0)
{
    void *const stream = (void *)torch_stream.stream();
    PreparedUse use;
    if (!cache_enabled() || !cacheable(rp) || !cacheable(ci) || trw_stream_is_capturing(stream)) return use;
    std::lock_guard<std::mutex> lock(g_cache_mu);
    std::shared_ptr<PreparedGraph> hit;
    for (auto it = g_cache.begin(); it != g_cache.end();) {
        if (entry_matches(**it, rp, ci, kind, pad)) {
            hit = *it;
            it = g_cache.erase(it);
        } else if ((*it)->rp_storage.expired() || (*it)->ci_storage.expired()) {
            it = g_cache.erase(it);  // its graph is gone
        } else if (entry_same_memory(**it, rp, ci, kind, pad)) {
            // the same tensors, written in place since: this entry can never match again -- it goes now,
            // pinned or not, so that its blob is free before the successor's is allocated
            it = g_cache.erase(it);
        } else {
            ++it;
        }
    }
    if (!hit) {
        hit = std::make_shared<PreparedGraph>(rp.storage().getWeakStorageImpl(), ci.storage().getWeakStorageImpl());
        hit->rp_ptr = rp.data_ptr();
        hit->ci_ptr = ci.data_ptr();
        hit->rp_numel = rp.numel();
        hit->ci_numel = ci.numel();
        hit->rp_version = rp._version();
        hit->ci_version = ci._version();
        hit->device = rp.device();
        hit->kind = kind;
        hit->pad = pad;
        hit->what = -100;  // nothing built yet
    }
    int32_t want = needed < 0 ? 0 : needed;
    if (hit->what >= 0) want |= hit->what;
    if (kind == 0) {
        // a graph that is walked again, and enough, is worth the masks of its longer rows (thresholds above); not behind
        // rw.prepare()'s back: that graph has what was asked
        const int64_t entries = ci.numel() > 0 ? ci.numel() : 1;
        if (biased_walk && (want & TRW_PREP_BIASED) && hit->biased_walks >= 1 && !hit->pinned &&
            hit->biased_steps / entries >= kMarksAfterStepsPerEntry)
            want |= TRW_PREP_MARKS;
        if (biased_walk && (want & TRW_PREP_MARKS) && !hit->pinned && hit->biased_steps / entries >= kFullMarksAfterStepsPerEntry)
            want |= TRW_PREP_MARKS_FULL;
        hit->biased_steps += biased_steps;
        // One uniform call is worth the 16-byte entries when it makes more steps than the graph has entries
        // (trw_prepare_plan); calls that are smaller than that each -- one GPU's shard of BASELINE configs[3] -- are
        // worth them together: from the call on with which the steps made over this graph reach its entry count.
        hit->uniform_steps += uniform_steps;
        const int64_t nnz = ci.numel();
        if (uniform_steps > 0 && !hit->pinned && hit->uniform_steps >= nnz && (rp.numel() + nnz) * 4 > (int64_t(32) << 20))
            want |= TRW_PREP_UNIFORM;
        want &= ~hit->refused;
        if (want & TRW_PREP_MARKS_FULL) want |= TRW_PREP_MARKS;
        if (want & TRW_PREP_MARKS) want |= TRW_PREP_BIASED;
    } else if (want & TRW_ELPREP_ENTRIES) {
        want |= TRW_ELPREP_TABLE;
    }
    want &= ~hit->refused;
    if (hit->what < 0 || (want & ~hit->what) != 0) build_blob(*hit, rp, ci, want, stream);
    if (biased_walk) ++hit->biased_walks;
    hit->pinned = hit->pinned || pin;
    g_cache.push_front(hit);
    {
        std::vector<int64_t> device_of;
        std::vector<bool> pinned;
        for (const auto &e : g_cache) {
            device_of.push_back((int64_t)e->device.index());
            pinned.push_back(e->pinned);
        }
        const auto victims = cache_victims(device_of, pinned, g_cache_capacity.load());  // (descending positions)
        for (const int64_t v : victims) g_cache.erase(std::next(g_cache.begin(), v));
    }
    if (hit->built_on != stream) {  // built on another stream: order after it, keep the blob alive for this one
        const int rc = trw_event_wait(hit->ready, stream);
        TORCH_CHECK(rc == TRW_OK, "torch_rw: ", trw_last_error());
        hit->blob.record_stream(torch_stream.unwrap());
    }
    use.blob = hit->blob;
    use.what = hit->what;
    use.valid = true;
    return use;
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
    // ... or NARROW rows: an int32 `out` takes the uint32 bit patterns of the node ids (the wire format of the multi-GPU
    // gather), written by the walk kernel itself where the production kernels run, by a narrowing pass over int64 rows
    // otherwise -- the same values either way.
    torch::Tensor walks;
    const bool narrow_out = out.has_value() && out->defined() && out->scalar_type() == torch::kInt32;
    if (out.has_value() && out->defined()) {
        walks = *out;
        TORCH_CHECK((walks.scalar_type() == torch::kInt64 || narrow_out) && walks.device() == rp.device() &&
                        walks.is_contiguous() && walks.dim() == 2 && walks.size(0) == n_walkers &&
                        walks.size(1) == walk_length + 1,
                    "out must be a contiguous int64 (or, for narrow rows, int32) [len(target_nodes), walk_length + 1] tensor "
                    "on the inputs' device");
        TORCH_CHECK(!narrow_out || n_nodes <= (int64_t(1) << 32), "narrow rows hold node ids below 2^32");
    } else {
        walks = torch::empty({n_walkers, walk_length + 1}, rp.options());
    }
    if (narrow_out && n_walkers == 0) return walks;

    void *const hs = (void *)stream.stream();
    // what a call of this shape is worth preparing; < 0: it runs the int64 kernels and wants no shadow at all
    const int32_t needed = trw_prepare_plan(n_nodes, nnz, n_walkers, (int32_t)walk_length, p, q, (int32_t)rng_mode);
    // (a serial replay rides a prepared graph too, but it is no sign that the graph is worth the full masks)
    const bool serial_replay = rng_mode == TRW_RNG_GLIBC ||
                               (rng_mode == TRW_RNG_AUTO && walker_base == 0 && n_walkers <= 32768 / (walk_length + 1));
    const bool biased_parallel = !(p == 1.0 && q == 1.0) && !serial_replay;
    PreparedUse prepared;
    // (a strided input was copied by .contiguous() above: a temporary is not worth a cache entry)
    // reference-exact uniform walks over the narrow layout: their steps count towards the 16-byte entries
    const bool uniform_exact = p == 1.0 && q == 1.0 && rng_mode != TRW_RNG_WALKER;
    if (n_walkers > 0 && needed >= 0 && rp.is_same(row_ptr) && ci.is_same(col_idx))
        prepared = prepared_for(rp, ci, needed, biased_parallel, false, stream, 0, 0,
                                uniform_exact ? n_walkers * walk_length : 0, biased_parallel ? n_walkers * walk_length : 0);

    // the walk over a prepared graph, into `walks`: narrow rows from the kernel where its family writes them, from a
    // narrowing pass over int64 rows where it does not (the serial replay, graphs on the int64 kernels, ...)
    torch::Tensor wide;
    auto run_walk = [&](const void *blob_ptr, size_t blob_bytes, int32_t what_built, void *ws_ptr, size_t ws_size) {
        int r = TRW_E_UNSUPPORTED;
        if (narrow_out)
            r = trw_walk_csr_prepared_u32_hip(blob_ptr, blob_bytes, what_built, rp.data_ptr<int64_t>(), n_nodes,
                                              ci.data_ptr<int64_t>(), nnz, tg.data_ptr<int64_t>(), n_walkers, walker_base, p, q,
                                              (int32_t)walk_length, (int32_t)seed, (int32_t)rng_mode,
                                              (uint32_t *)walks.data_ptr<int32_t>(), ws_ptr, ws_size, hs);
        if (r != TRW_E_UNSUPPORTED) return r;
        if (narrow_out) wide = torch::empty({n_walkers, walk_length + 1}, rp.options());
        r = trw_walk_csr_prepared_hip(blob_ptr, blob_bytes, what_built, rp.data_ptr<int64_t>(), n_nodes, ci.data_ptr<int64_t>(),
                                      nnz, tg.data_ptr<int64_t>(), n_walkers, walker_base, p, q, (int32_t)walk_length,
                                      (int32_t)seed, (int32_t)rng_mode,
                                      narrow_out ? wide.data_ptr<int64_t>() : walks.data_ptr<int64_t>(), ws_ptr, ws_size, hs);
        if (r == TRW_OK && narrow_out)
            r = trw_rows_narrow_hip(wide.data_ptr<int64_t>(), wide.numel(), (uint32_t *)walks.data_ptr<int32_t>(), nullptr, hs);
        return r;
    };
    torch::Tensor workspace;
    int rc;
    if (prepared.valid) {
        const size_t ws_bytes =
            trw_walk_csr_prepared_workspace_bytes(n_walkers, (int32_t)walk_length, p, q, (int32_t)rng_mode);
        workspace = torch::empty({(int64_t)ws_bytes}, rp.options().dtype(torch::kUInt8));
        pybind11::gil_scoped_release nogil;
        rc = run_walk(prepared.blob.data_ptr(), (size_t)prepared.blob.numel(), prepared.what, workspace.data_ptr(), ws_bytes);
    } else {
        // self-contained call (cache off, capture in progress, temporaries): prepare into this call's own scratch.
        // When the device is short of memory the prepared part is retried leaner -- the edge table and the fat entries
        // are optional, the walk then takes the kernel that does without (the reference needs no scratch at all).
        const size_t call_bytes =
            trw_walk_csr_prepared_workspace_bytes(n_walkers, (int32_t)walk_length, p, q, (int32_t)rng_mode);
        const int32_t ladder[3] = {needed, needed > 0 ? 0 : needed, needed >= 0 ? TRW_PREP_ORDER_ONLY : needed};
        torch::Tensor blob, scratch;
        int32_t what = needed;
        size_t scratch_bytes = 0;
        for (int k = 0; k < 3; ++k) {
            if (k > 0 && ladder[k] == ladder[k - 1]) continue;
            what = ladder[k];
            try {
                blob = torch::empty({(int64_t)trw_prepare_csr_persistent_bytes(n_nodes, nnz, what)}, rp.options().dtype(torch::kUInt8));
                scratch_bytes = trw_prepare_csr_scratch_bytes(n_nodes, nnz, what);
                if (scratch_bytes) scratch = torch::empty({(int64_t)scratch_bytes}, rp.options().dtype(torch::kUInt8));
                break;
            } catch (const c10::OutOfMemoryError &) {
                if (k == 2) throw;
            }
        }
        workspace = torch::empty({(int64_t)call_bytes}, rp.options().dtype(torch::kUInt8));
        pybind11::gil_scoped_release nogil;
        rc = trw_prepare_csr_scratch_hip(rp.data_ptr<int64_t>(), n_nodes, ci.data_ptr<int64_t>(), nnz, what, blob.data_ptr(),
                                         (size_t)blob.numel(), scratch_bytes ? scratch.data_ptr() : nullptr, scratch_bytes, hs);
        if (rc == TRW_OK) rc = run_walk(blob.data_ptr(), (size_t)blob.numel(), what, workspace.data_ptr(), call_bytes);
    }
    TORCH_CHECK(rc == TRW_OK, "torch_rw walk failed (", rc, "): ", trw_last_error());
    finish_status(workspace, hs, check_mode, "walk");
    return walks;
}

// ---- wire format of the multi-GPU gather (include/trw.h: trw_rows_narrow / trw_rows_widen) ---------------
// rows: contiguous int64, any shape; wire: int32 of the same shape holding the uint32 bit patterns (torch's
// collectives know int32).  `strict`: synchronise and raise when a value did not fit (node ids >= 2^32).
torch::Tensor narrow_rows(const torch::Tensor &rows, const c10::optional<torch::Tensor> &out, bool strict)
{
    TORCH_CHECK(rows.scalar_type() == torch::kInt64 && rows.is_contiguous(), "rows must be a contiguous int64 tensor");
    torch::Tensor wire;
    if (out.has_value() && out->defined()) {
        wire = *out;
        TORCH_CHECK(wire.scalar_type() == torch::kInt32 && wire.is_contiguous() && wire.device() == rows.device() &&
                        wire.numel() == rows.numel(), "out must be a contiguous int32 tensor of rows' size and device");
    } else {
        wire = torch::empty(rows.sizes(), rows.options().dtype(torch::kInt32));
    }
    int rc;
    if (!rows.is_cuda()) {
        int64_t overflow = 0;
        rc = trw_rows_narrow_cpu(rows.data_ptr<int64_t>(), rows.numel(), (uint32_t *)wire.data_ptr<int32_t>(), &overflow);
        TORCH_CHECK(rc == TRW_OK, "torch_rw narrow_rows failed (", rc, "): ", trw_last_error());
        TORCH_CHECK(!strict || overflow == 0, "narrow_rows: a value does not fit the uint32 wire format");
        return wire;
    }
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(rows.device());
    const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(rows.device().index());
    torch::Tensor flag;
    if (strict) flag = torch::zeros({1}, rows.options());
    rc = trw_rows_narrow_hip(rows.data_ptr<int64_t>(), rows.numel(), (uint32_t *)wire.data_ptr<int32_t>(),
                             strict ? flag.data_ptr<int64_t>() : nullptr, (void *)stream.stream());
    TORCH_CHECK(rc == TRW_OK, "torch_rw narrow_rows failed (", rc, "): ", trw_last_error());
    if (strict) TORCH_CHECK(flag.item<int64_t>() == 0, "narrow_rows: a value does not fit the uint32 wire format");
    return wire;
}

torch::Tensor widen_rows(const torch::Tensor &wire, const c10::optional<torch::Tensor> &out)
{
    TORCH_CHECK(wire.scalar_type() == torch::kInt32 && wire.is_contiguous(), "wire must be a contiguous int32 tensor");
    torch::Tensor rows;
    if (out.has_value() && out->defined()) {
        rows = *out;
        TORCH_CHECK(rows.scalar_type() == torch::kInt64 && rows.is_contiguous() && rows.device() == wire.device() &&
                        rows.numel() == wire.numel(), "out must be a contiguous int64 tensor of wire's size and device");
    } else {
        rows = torch::empty(wire.sizes(), wire.options().dtype(torch::kInt64));
    }
    int rc;
    if (!wire.is_cuda()) {
        rc = trw_rows_widen_cpu((const uint32_t *)wire.data_ptr<int32_t>(), wire.numel(), rows.data_ptr<int64_t>());
    } else {
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(wire.device());
        const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(wire.device().index());
        rc = trw_rows_widen_hip((const uint32_t *)wire.data_ptr<int32_t>(), wire.numel(), rows.data_ptr<int64_t>(),
                                (void *)stream.stream());
    }
    TORCH_CHECK(rc == TRW_OK, "torch_rw widen_rows failed (", rc, "): ", trw_last_error());
    return rows;
}

// rw.prepare(): build (or upgrade) the cached prepared graph now; returns the TRW_PREP_* bits it holds.
int64_t prepare(const torch::Tensor &row_ptr, const torch::Tensor &col_idx, int64_t what)
{
    check_index_tensor(row_ptr, "row_ptr");
    check_index_tensor(col_idx, "col_idx");
    TORCH_CHECK(row_ptr.device() == col_idx.device(), "row_ptr and col_idx must be on the same device");
    TORCH_CHECK(row_ptr.numel() >= 1, "row_ptr must have num_nodes + 1 entries");
    TORCH_CHECK(row_ptr.is_contiguous() && col_idx.is_contiguous(),
                "prepare() keys on the tensors' storage: row_ptr and col_idx must be contiguous");
    TORCH_CHECK(what >= 0 && what <= (TRW_PREP_UNIFORM | TRW_PREP_BIASED | TRW_PREP_MARKS | TRW_PREP_MARKS_FULL),
                "unknown prepare flags ", what);
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(row_ptr.device());
    const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(row_ptr.device().index());
    TORCH_CHECK(cache_enabled(), "the prepared-graph cache is switched off (TRW_CACHE=0 / set_cache(False))");
    const auto entry = prepared_for(row_ptr, col_idx, (int32_t)what, false, true, stream);
    TORCH_CHECK(entry.valid, "this graph cannot be cached (inference tensors, or the stream is being captured)");
    return (int64_t)entry.what;
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
    const bool on_gpu = triples_indexed.is_cuda();  // rw_init.cpp:57: the device of the first tensor picks the leg
    const int check_mode = resolve_check(check);
    if (on_gpu && check_mode == kCheckDeferred) raise_pending();
    check_index_tensor(target_nodes, "target_nodes", on_gpu);
    for (const auto *t : {&triples_indexed, &relation_tail_index}) {
        TORCH_CHECK(t->is_cuda() == on_gpu, "triples_indexed, relation_tail_index and target_nodes must be on the same device");
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
    if (!on_gpu) {  // the host leg (rw_init.cpp:66-74 -> triples::walk_triples_cpu): csrc/cpu on one thread
        const auto tr = triples_indexed.contiguous();
        const auto ix = relation_tail_index.contiguous();
        const auto tg = target_nodes.contiguous();
        auto walks = torch::empty({tg.numel(), 2 * walk_length + 1}, tg.options());
        int rc;
        {
            pybind11::gil_scoped_release nogil;
            rc = trw_walk_triples_cpu(tr.data_ptr<int64_t>(), tr.size(0), ix.data_ptr<int64_t>(), ix.size(0),
                                      tg.data_ptr<int64_t>(), tg.numel(), (int32_t)walk_length, padding_idx,
                                      (int32_t)seed, walks.data_ptr<int64_t>());
        }
        TORCH_CHECK(rc == TRW_OK, "torch_rw walk_triples failed (", rc, "): ", trw_last_error());
        return walks;
    }

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
    const bool on_gpu = edge_list_indexed.is_cuda();  // rw_init.cpp:39
    const int check_mode = resolve_check(check);
    if (on_gpu && check_mode == kCheckDeferred) raise_pending();
    check_index_tensor(target_nodes, "target_nodes", on_gpu);
    for (const auto *t : {&edge_list_indexed, &node_edges_idx}) {
        TORCH_CHECK(t->is_cuda() == on_gpu, "edge_list_indexed, node_edge_index and target_nodes must be on the same device");
        TORCH_CHECK(t->scalar_type() == torch::kInt64, "expected scalar type Long but found ",
                    c10::toString(t->scalar_type()));
        TORCH_CHECK(t->dim() == 2 && t->size(1) == 2, "edge_list_indexed / node_edge_index must be [*, 2]");
    }
    TORCH_CHECK(edge_list_indexed.device() == target_nodes.device() && node_edges_idx.device() == target_nodes.device(),
                "edge_list_indexed, node_edge_index and target_nodes must be on the same device");
    TORCH_CHECK(walk_length >= 1, "walk_length must be >= 1, got ", walk_length);
    TORCH_CHECK(walk_length < (int64_t(1) << 31) && seed >= INT32_MIN && seed <= INT32_MAX,
                "walk_length/seed must fit the reference's int arguments");
    if (!on_gpu) {  // the host leg (rw_init.cpp:41-44 -> walk_edge_list_cpu): csrc/cpu on one thread
        const auto ed = edge_list_indexed.contiguous();
        const auto ix = node_edges_idx.contiguous();
        const auto tg = target_nodes.contiguous();
        auto walks = torch::empty({tg.numel(), walk_length + 1}, tg.options());
        int rc;
        {
            pybind11::gil_scoped_release nogil;
            rc = trw_walk_edge_list_cpu(ed.data_ptr<int64_t>(), ed.size(0), ix.data_ptr<int64_t>(), ix.size(0),
                                        tg.data_ptr<int64_t>(), tg.numel(), p, q, (int32_t)walk_length, (int32_t)seed,
                                        padding_idx, restart ? 1 : 0, walks.data_ptr<int64_t>());
        }
        TORCH_CHECK(rc == TRW_OK, "torch_rw walk_edge_list failed (", rc, "): ", trw_last_error());
        return walks;
    }

    const c10::hip::HIPGuardMasqueradingAsCUDA guard(target_nodes.device());
    const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(target_nodes.device().index());
    const auto ed = edge_list_indexed.contiguous();
    const auto ix = node_edges_idx.contiguous();
    const auto tg = target_nodes.contiguous();
    const int64_t n_edges = ed.size(0), n_nodes = ix.size(0), n_walkers = tg.numel();
    auto walks = torch::empty({n_walkers, walk_length + 1}, tg.options());
    void *const hs = (void *)stream.stream();
    const int32_t needed = trw_prepare_edge_list_plan(n_edges, n_nodes, n_walkers, (int32_t)walk_length, p, q, padding_idx,
                                                      (int32_t)rng_mode);
    PreparedUse prepared;
    if (n_walkers > 0 && needed >= 0 && ed.is_same(edge_list_indexed) && ix.is_same(node_edges_idx))
        prepared = prepared_for(ed, ix, needed, false, false, stream, 1, padding_idx);
    torch::Tensor workspace;
    int rc;
    if (prepared.valid) {
        workspace = torch::empty({256}, tg.options().dtype(torch::kUInt8));
        pybind11::gil_scoped_release nogil;
        rc = trw_walk_edge_list_prepared_hip(prepared.blob.data_ptr(), (size_t)prepared.blob.numel(), prepared.what,
                                             ed.data_ptr<int64_t>(), n_edges, ix.data_ptr<int64_t>(), n_nodes,
                                             tg.data_ptr<int64_t>(), n_walkers, walker_base, p, q, (int32_t)walk_length,
                                             (int32_t)seed, padding_idx, restart ? 1 : 0, (int32_t)rng_mode,
                                             walks.data_ptr<int64_t>(), workspace.data_ptr(), 256, hs);
    } else {
        const size_t ws_bytes = trw_walk_edge_list_workspace_bytes(n_edges, n_nodes, n_walkers, (int32_t)walk_length,
                                                                   padding_idx, (int32_t)rng_mode);
        workspace = torch::empty({(int64_t)ws_bytes}, tg.options().dtype(torch::kUInt8));
        pybind11::gil_scoped_release nogil;
        rc = trw_walk_edge_list_hip(ed.data_ptr<int64_t>(), n_edges, ix.data_ptr<int64_t>(), n_nodes,
                                    tg.data_ptr<int64_t>(), n_walkers, walker_base, p, q, (int32_t)walk_length,
                                    (int32_t)seed, padding_idx, restart ? 1 : 0, (int32_t)rng_mode,
                                    walks.data_ptr<int64_t>(), workspace.data_ptr(), ws_bytes, hs);
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
    TORCH_CHECK(walks.scalar_type() == torch::kInt64, "expected scalar type Long but found ",
                c10::toString(walks.scalar_type()));
    TORCH_CHECK(walks.dim() == 2, "walks must be 2-dimensional");
    TORCH_CHECK(walks.is_contiguous(), "walks must be a contigous tensor");  // windows_cpu.cpp:15 (sic)
    const int64_t num_walks = walks.size(0), cols = walks.size(1);
    TORCH_CHECK(window_size >= 1 && window_size <= cols, "window_size must be between 1 and the walk width");
    TORCH_CHECK(seed >= INT32_MIN && seed <= INT32_MAX, "seed must fit the reference's int argument");
    if (!walks.is_cuda()) {  // the host leg (rw_init.cpp:85-87 / :98-100 -> to_windows{,_cbow}_cpu)
        const int64_t Wc = (cols - window_size + 1) * num_walks;
        auto centre = torch::empty({Wc}, walks.options());
        auto rest = torch::empty({Wc, window_size - 1}, walks.options());
        auto neg = cbow ? torch::empty({Wc}, walks.options()) : torch::empty({Wc, window_size - 1}, walks.options());
        int rc;
        {
            pybind11::gil_scoped_release nogil;
            rc = trw_to_windows_cpu(walks.data_ptr<int64_t>(), num_walks, cols, (int32_t)window_size, num_nodes,
                                    (int32_t)seed, cbow ? 1 : 0, centre.data_ptr<int64_t>(), rest.data_ptr<int64_t>(),
                                    neg.data_ptr<int64_t>());
        }
        TORCH_CHECK(rc == TRW_OK, "torch_rw to_windows failed (", rc, "): ", trw_last_error());
        if (cbow) return std::make_tuple(centre, neg, rest);
        return std::make_tuple(centre, rest, neg);
    }
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

// rw.walk_windows: walk followed by to_windows (torch_rw/rw.py:3-4, :29-30) as one call.  Returns (target_nodes,
// pos_windows, neg_windows, walks-or-empty), equal to to_windows(walk_ex(...), window_size, num_nodes, window_seed)
// with the windows numbered from walker_base.  The reference-exact uniform walk over a cached prepared graph runs as
// ONE walk kernel that writes the windows from its LDS tile (include/trw.h: trw_walk_windows_csr_prepared_hip); every
// other case (node2vec walks, CPU tensors, window_size > 9, no cache) is the two calls, composed here.
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
walk_windows(const torch::Tensor &row_ptr, const torch::Tensor &col_idx, const torch::Tensor &target_nodes, double p,
             double q, int64_t walk_length, int64_t seed, int64_t window_size, int64_t num_nodes, int64_t window_seed,
             int64_t walker_base, bool return_walks, int64_t check)
{
    TORCH_CHECK(walk_length >= 1, "walk_length must be >= 1, got ", walk_length);
    TORCH_CHECK(window_size >= 1 && window_size <= walk_length + 1, "window_size must be between 1 and the walk width");
    TORCH_CHECK(seed >= INT32_MIN && seed <= INT32_MAX && window_seed >= INT32_MIN && window_seed <= INT32_MAX &&
                    walk_length < (int64_t(1) << 31), "walk_length/seed must fit the reference's int arguments");
    const bool uniform = p == 1.0 && q == 1.0;
    // the walk's stream is resolved exactly as rw.walk resolves it (process-wide mode; AUTO -> the reference's libc
    // stream for a uniform walk), so that "what the two calls give" holds under set_rng_mode(RNG_WALKER) too (ADVICE
    // r03): the fused kernel IS the reference-exact uniform walk, any other stream runs as the two calls
    const int64_t rng = (int64_t)g_rng_mode.load();
    const bool glibc_uniform = uniform && (rng == TRW_RNG_AUTO || rng == TRW_RNG_GLIBC);
    auto composed = [&]() {
        auto walks = walk_ex(row_ptr, col_idx, target_nodes, p, q, walk_length, seed, walker_base, rng, check, c10::nullopt);
        auto w = windows_impl(walks, window_size, num_nodes, window_seed, false, walker_base, TRW_RNG_GLIBC);
        return std::make_tuple(std::get<0>(w), std::get<1>(w), std::get<2>(w), return_walks ? walks : torch::Tensor());
    };
    if (!row_ptr.is_cuda() || !glibc_uniform) return composed();
    check_index_tensor(row_ptr, "row_ptr");
    check_index_tensor(col_idx, "col_idx");
    check_index_tensor(target_nodes, "target_nodes");
    TORCH_CHECK(row_ptr.device() == col_idx.device() && row_ptr.device() == target_nodes.device(),
                "row_ptr, col_idx and target_nodes must be on the same device");
    TORCH_CHECK(row_ptr.numel() >= 1, "row_ptr must have num_nodes + 1 entries");
    const int64_t n_nodes = row_ptr.numel() - 1, nnz = col_idx.numel(), n_walkers = target_nodes.numel();
    if (n_walkers == 0 || !row_ptr.is_contiguous() || !col_idx.is_contiguous()) return composed();
    const int check_mode = resolve_check(check);
    if (check_mode == kCheckDeferred) raise_pending();
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(row_ptr.device());
    const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(row_ptr.device().index());
    const auto tg = target_nodes.contiguous();
    const int32_t needed = trw_prepare_plan(n_nodes, nnz, n_walkers, (int32_t)walk_length, 1.0, 1.0, TRW_RNG_GLIBC);
    PreparedUse prepared;
    if (needed >= 0) prepared = prepared_for(row_ptr, col_idx, needed, false, false, stream, 0, 0, n_walkers * walk_length, 0);
    if (!prepared.valid) return composed();
    const bool fused = trw_walk_windows_fused(n_nodes, nnz, n_walkers, (int32_t)walk_length, (int32_t)window_size,
                                              prepared.what) == 1;
    const int64_t W = (walk_length + 1 - window_size + 1) * n_walkers;
    auto opts = row_ptr.options();
    auto centre = torch::empty({W}, opts);
    auto rest = torch::empty({W, window_size - 1}, opts);
    auto neg = torch::empty({W, window_size - 1}, opts);
    torch::Tensor walks;
    if (return_walks || !fused) walks = torch::empty({n_walkers, walk_length + 1}, opts);
    const size_t ws_bytes = trw_walk_windows_workspace_bytes(n_walkers, (int32_t)walk_length, (int32_t)window_size);
    auto workspace = torch::empty({(int64_t)ws_bytes}, opts.dtype(torch::kUInt8));
    void *const hs = (void *)stream.stream();
    int rc;
    {
        pybind11::gil_scoped_release nogil;
        rc = trw_walk_windows_csr_prepared_hip(prepared.blob.data_ptr(), (size_t)prepared.blob.numel(), prepared.what,
                                               row_ptr.data_ptr<int64_t>(), n_nodes, col_idx.data_ptr<int64_t>(), nnz,
                                               tg.data_ptr<int64_t>(), n_walkers, walker_base, (int32_t)walk_length,
                                               (int32_t)seed, (int32_t)window_size, num_nodes, (int32_t)window_seed,
                                               walks.defined() ? walks.data_ptr<int64_t>() : nullptr,
                                               centre.data_ptr<int64_t>(), rest.data_ptr<int64_t>(),
                                               neg.data_ptr<int64_t>(), workspace.data_ptr(), ws_bytes, hs);
    }
    TORCH_CHECK(rc == TRW_OK, "torch_rw walk_windows failed (", rc, "): ", trw_last_error());
    finish_status(workspace, hs, check_mode, "walk_windows");
    return std::make_tuple(centre, rest, neg, return_walks ? walks : torch::Tensor());
}

// to_windows_triples / to_windows_triples_cbow: reference dispatchers csrc/rw_init.cpp:103-131.
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> triples_windows_impl(const torch::Tensor &walks,
                                                                             int64_t window_size, int64_t padding_idx,
                                                                             const torch::Tensor &triples, int64_t seed,
                                                                             bool cbow, int64_t base, int64_t rng_mode)
{
    for (const auto *t : {&walks, &triples}) {
        TORCH_CHECK(t->scalar_type() == torch::kInt64, "expected scalar type Long but found ",
                    c10::toString(t->scalar_type()));
        TORCH_CHECK(t->dim() == 2, "walks and triples must be 2-dimensional");
    }
    TORCH_CHECK(walks.is_contiguous(), "walks must be a contigous tensor");  // windows_cpu.cpp:172 (sic)
    TORCH_CHECK(triples.size(1) == 3 && triples.size(0) >= 1, "triples must be [T >= 1, 3]");
    TORCH_CHECK(walks.device() == triples.device(), "walks and triples must be on the same device");
    TORCH_CHECK(window_size >= 1, "window_size must be >= 1");
    TORCH_CHECK(seed >= INT32_MIN && seed <= INT32_MAX, "seed must fit the reference's int argument");
    if (!walks.is_cuda()) {  // the host leg (rw_init.cpp:113-115 / :128-130 -> to_windows_triples{,_cbow}_cpu)
        const auto trc = triples.contiguous();
        const int64_t nw = walks.size(0), cw = walks.size(1), Wc = ((cw - 1) / 2) * nw;
        auto centre = torch::empty({Wc, 3}, walks.options());
        auto pos = torch::empty({Wc, 2 * window_size, 3}, walks.options());
        auto neg = cbow ? torch::empty({Wc, 3}, walks.options()) : torch::empty({Wc, 2 * window_size, 3}, walks.options());
        int rc;
        {
            pybind11::gil_scoped_release nogil;
            rc = trw_to_windows_triples_cpu(walks.data_ptr<int64_t>(), nw, cw, (int32_t)window_size, padding_idx,
                                            trc.data_ptr<int64_t>(), trc.size(0), (int32_t)seed, cbow ? 1 : 0,
                                            centre.data_ptr<int64_t>(), pos.data_ptr<int64_t>(), neg.data_ptr<int64_t>());
        }
        TORCH_CHECK(rc == TRW_OK, "torch_rw to_windows_triples failed (", rc, "): ", trw_last_error());
        if (cbow) return std::make_tuple(centre, neg, pos);
        return std::make_tuple(centre, pos, neg);
    }
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
    m.def("walk_windows", &walk_windows, "walk + to_windows in one call (the uniform walk: one fused kernel)",
          pybind11::arg("row_ptr"), pybind11::arg("col_idx"), pybind11::arg("target_nodes"), pybind11::arg("p"),
          pybind11::arg("q"), pybind11::arg("walk_length"), pybind11::arg("seed"), pybind11::arg("window_size"),
          pybind11::arg("num_nodes"), pybind11::arg("window_seed"), pybind11::arg("walker_base") = (int64_t)0,
          pybind11::arg("return_walks") = false, pybind11::arg("check") = (int64_t)-1);
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
    m.def("prepare", &prepare, "build the cached prepared graph of (row_ptr, col_idx) now", pybind11::arg("row_ptr"),
          pybind11::arg("col_idx"), pybind11::arg("what"));
    m.def("set_cache", [](bool on, int64_t capacity) {
        g_cache_on.store(on ? 1 : 0);
        if (capacity > 0) g_cache_capacity.store(capacity);
        if (!on) {
            std::lock_guard<std::mutex> lock(g_cache_mu);
            g_cache.clear();
        }
    }, pybind11::arg("on"), pybind11::arg("capacity") = (int64_t)0);
    m.def("narrow_rows", &narrow_rows, "int64 rows -> the uint32 wire format of the multi-GPU gather (as int32)",
          pybind11::arg("rows"), pybind11::arg("out") = c10::optional<torch::Tensor>(), pybind11::arg("strict") = false);
    m.def("widen_rows", &widen_rows, "uint32 wire format (as int32) -> int64 rows", pybind11::arg("wire"),
          pybind11::arg("out") = c10::optional<torch::Tensor>());
    m.def("_cache_victims", &cache_victims,
          "the prepared-graph cache's eviction rule (positions to drop; capacity counts per device)",
          pybind11::arg("device_of"), pybind11::arg("pinned"), pybind11::arg("capacity"));
    m.def("invalidate", [](const torch::Tensor &a, const torch::Tensor &b) {
        // Drop what is cached for these tensors' memory, whatever their version counters say: for writes torch cannot
        // see (through `.data`, DLPack, another extension's kernels), which leave the counters where they were.
        std::lock_guard<std::mutex> lock(g_cache_mu);
        int64_t dropped = 0;
        for (auto it = g_cache.begin(); it != g_cache.end();) {
            if (a.has_storage() && b.has_storage() && entry_same_memory(**it, a, b, (*it)->kind, (*it)->pad)) {
                it = g_cache.erase(it);
                ++dropped;
            } else {
                ++it;
            }
        }
        return dropped;
    }, "forget the prepared graph cached for (row_ptr, col_idx) / (edge_list_indexed, node_edge_index)");
    m.def("clear_cache", []() {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        g_cache.clear();
    });
    m.def("prepare_timing", []() {
        pybind11::dict d;
        d["alloc_ms"] = g_last_alloc_ms.load();
        d["launch_ms"] = g_last_launch_ms.load();
        d["prepared_bytes"] = g_last_keep_bytes.load();
        d["scratch_bytes"] = g_last_scratch_bytes.load();
        return d;
    }, "host-side cost of the last prepared-graph build: allocator, kernel launches; and its two sizes");
    m.def("cache_info", []() {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        std::vector<std::tuple<int64_t, int64_t, int64_t, int64_t, int64_t>> out;
        for (const auto &e : g_cache)
            out.emplace_back(e->kind == 1 ? e->ci_numel / 2 : e->rp_numel - 1, e->kind == 1 ? e->rp_numel / 2 : e->ci_numel,
                             (int64_t)e->what, (int64_t)e->blob.numel(), e->biased_walks);
        return out;
    });
    m.attr("PREP_UNIFORM") = (int64_t)TRW_PREP_UNIFORM;
    m.attr("PREP_BIASED") = (int64_t)TRW_PREP_BIASED;
    m.attr("PREP_MARKS") = (int64_t)TRW_PREP_MARKS;
    m.attr("PREP_MARKS_FULL") = (int64_t)TRW_PREP_MARKS_FULL;
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
