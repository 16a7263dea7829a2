// C ABI (include/emagls.h) and host-side orchestration of the design pipelines.
// The host code only sequences launches and derives scalar constants (nfft, k_cut, simulation
// order: lib/getEMagLsFilters.m:44-48, dependencies/getSMAIRMatrix.m:95); all array arithmetic runs
// in the HIP kernels.  There is no CPU fallback: without a GPU every entry point returns an error.
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/emagls.h"
#include "kernels.hpp"

using namespace emagls;

namespace {

thread_local std::string g_last_error;
// EMAGLS_JOBS_TRACE=1: wall-clock marks of the job lists' host-side phases on stderr
static bool trace_on() { static const bool t = getenv("EMAGLS_JOBS_TRACE") != nullptr; return t; }
static void trace_mark(const char* what) {
    if (!trace_on()) return;
    static const auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "emagls trace: %-44s %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}

constexpr double C_SOUND = 343.0;       // dependencies/getSMAIRMatrix.m:86
constexpr int NFFT_MAX_LEN = 2048;      // lib/getEMagLsFilters.m:35
constexpr double F_CUT_MIN_FREQ = 1e3;  // :36
constexpr double SVD_REGUL_CONST = 0.01;  // :39
constexpr int NFLAG = 8;                   // device-side status words of a design (plan_recover)
constexpr int SMAIR_DEFAULT_ORDER = 4;    // dependencies/getSMAIRMatrix.m:39-41 (params.order when the caller leaves it unset)

// Streams are recycled through a process-wide pool and never destroyed.  A design plan owns three and a long session creates
// and drops hundreds of plans (one-shot cache evictions, radius sweeps).  Under the HIP 7.0 runtime that torch bundles, a
// multi-stream graph capture on stream handles the runtime had recycled after many hipStreamDestroy calls produced a graph
// whose hipGraphLaunch dereferenced a null pointer (reproduced: tests/test_gpu_config4.py followed by test_gpu_parity.py, crash
// in the third custom-basis one-shot call; gone with the pool, and gone with single-stream capture).
struct StreamPool {
    std::mutex mu;
    std::map<int, std::vector<hipStream_t>> idle;   // per device
    static StreamPool& get() { static StreamPool* p = new StreamPool; return *p; }   // (never destroyed: outlives every plan)
    static bool enabled() { static const bool on = [] { const char* e = getenv("EMAGLS_STREAM_POOL"); return !(e && e[0] == '0'); }(); return on; }
    hipStream_t take() {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        {   // The first streams a process creates each open a hardware queue of their own (GPU_MAX_HW_QUEUES = 4), later ones share
            // those queues.  Since round 6 a job chunk's plans no longer take four streams each (hipStreamCreate was 3 ms of a plan's
            // set-up), so a lone chunk's batch would fork its stages onto exactly those first streams -- and ran 8 % slower that way
            // (2400-2470 against 2590-2670 sets/s at 20 steps, A/B on one box, profiles/r06_stream_warm.md; any number of parked streams
            // from 4 to 80 restores it).  So the pool parks 8 streams before it hands the first one out (EMAGLS_STREAM_WARM=n; 0: none).
            static const int warm = [] { const char* e = getenv("EMAGLS_STREAM_WARM"); return e ? atoi(e) : 8; }();
            static std::once_flag once;
            if (warm > 0) std::call_once(once, [&] {
                std::lock_guard<std::mutex> lk(mu);
                for (int i = 0; i < warm; ++i) { hipStream_t st = nullptr; if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) idle[dev].push_back(st); }
            });
        }
        if (enabled()) {
            std::lock_guard<std::mutex> lk(mu);
            auto& v = idle[dev];
            if (!v.empty()) { hipStream_t st = v.back(); v.pop_back(); return st; }
        }
        hipStream_t st = nullptr;
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        return st;
    }
    void give(hipStream_t st) {
        if (!st) return;
        if (!enabled()) { hipStreamDestroy(st); return; }
        hipStreamSynchronize(st);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { hipStreamDestroy(st); return; }
        std::lock_guard<std::mutex> lk(mu);
        idle[dev].push_back(st);
    }
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool owned = false;   // allocated on its own (hipFree when dropped); false once a batch moved it into its arena
};
// Device memory of plans (slabs) and batches (arenas) comes from a process-wide pool of blocks that are handed back instead of freed:
// a job list whose plans live for one chunk each otherwise pays hipMalloc / hipFree of ~0.35 GB per design again and again (and
// the calls were erratic next to running kernels: 30 ms ... 1.6 s for the plans of one chunk).  emagls_cache_clear() frees the pool;
// EMAGLS_POOL_GB (default 128) bounds what it keeps.
struct BlockPool {
    std::mutex mu;
    std::map<int, std::multimap<size_t, void*>> free_;   // device -> size -> block
    size_t held = 0;
    static BlockPool& get() { static BlockPool* p = new BlockPool; return *p; }   // (never destroyed: plans of static caches hand their blocks back at exit)
    static size_t cap() {
        // (128 GB of the 288: a rank's share of BASELINE config 4 holds two arenas of 21 GB -- the small radii keep materialised
        // operands, 1.5 GB per design -- plus as much again in released plan slabs while the next chunks are being built; with 64 GB the
        // arenas were freed and every list of new radii allocated them afresh, 0.3 ms ... 5 s per hipMalloc)
        static const size_t c = [] { const char* e = getenv("EMAGLS_POOL_GB"); return (size_t)(e ? std::max(0, atoi(e)) : 128) << 30; }();
        return c;
    }
    // Sizes of large blocks (batch arenas: gigabytes) come in classes -- multiples of an eighth of the power of two below them -- so that
    // the arenas of similar chunks (other array radii: routes, hence buffer sizes, a few per cent apart) are the SAME size and one
    // chunk's released arena serves the next exactly.  Fresh device memory is what a new chunk must not need: hipMalloc of a 4 GB arena
    // took 0.3 ms on one box and 0.5 ... 2.9 s next to running kernels on others (profiles/r06_cold_path.md).
    static size_t size_class(size_t bytes) {
        size_t step = (size_t)64 << 20;
        while (step * 16 <= bytes) step *= 2;
        return (bytes + step - 1) / step * step;
    }
    // a block of at least `bytes` (exactly `bytes` when it has to be allocated); *got = its size
    // (alloc_bytes: what a miss allocates -- an arena asks for a block that holds its need and, when there is none, allocates the size
    // class of an eighth more: the next list's need, a few per cent larger, then fits the block this one hands back)
    void* take(size_t bytes, size_t* got, size_t alloc_bytes = 0) {
        if (alloc_bytes < bytes) alloc_bytes = bytes;
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        {
            std::lock_guard<std::mutex> lk(mu);
            auto& fl = free_[dev];
            auto it = fl.lower_bound(bytes);
            // (the smallest block that is large enough, up to a quarter larger -- half larger for the gigabyte-sized arenas, whose need
            // moves by a few per cent from one list of array radii to the next: fresh device memory for 2 x 4.5 GB took 3.6 s there)
            if (it != fl.end() && it->first <= bytes + (bytes >= ((size_t)1 << 30) ? bytes / 2 : bytes / 4)) {
                void* p = it->second;
                *got = it->first;
                held -= it->first;
                fl.erase(it);
                return p;
            }
        }
        void* p = nullptr;
        const auto t_alloc0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&p, alloc_bytes);
        if (trace_on() && alloc_bytes >= ((size_t)256 << 20)) {
            std::lock_guard<std::mutex> lk(mu);
            std::string have;
            for (auto& kv : free_[dev]) if (kv.first >= ((size_t)256 << 20)) have += " " + std::to_string(kv.first >> 20);
            fprintf(stderr, "emagls trace: block pool miss: need %zu MB, hipMalloc of %zu MB took %.1f ms; large blocks in the pool (MB):%s\n", bytes >> 20, alloc_bytes >> 20,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc0).count(), have.c_str());
        }
        if (e == hipErrorOutOfMemory) {   // the pool may hold gigabytes of blocks of other sizes: return them to the runtime and try once more
            (void)hipGetLastError();
            clear();
            e = hipMalloc(&p, alloc_bytes);
        }
        HIP_CHECK(e);
        *got = alloc_bytes;
        return p;
    }
    void give(void* p, size_t bytes) {
        if (!p) return;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); hipFree(p); return; }
        std::lock_guard<std::mutex> lk(mu);
        if (held + bytes > cap()) { hipFree(p); return; }
        free_[dev].emplace(bytes, p);
        held += bytes;
    }
    void clear() {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& d : free_) for (auto& kv : d.second) hipFree(kv.second);
        free_.clear();
        held = 0;
    }
};

// one allocation that holds the buffers of all plans of a batch at a constant stride (see emagls_batch)
struct Arena {
    void* base = nullptr;
    size_t bytes = 0;
    ~Arena() { if (base) BlockPool::get().give(base, bytes); }
};

int round_up(int64_t v, int64_t m) { return (int)(ceil_div(v, m) * m); }

}  // namespace

struct emagls_batch;
void emagls_batch_forget(emagls_batch* b, emagls_plan* p);

struct emagls_plan {
    emagls_design_desc d{};
    int device = -1;              // the HIP device the plan was created on: every C entry point runs on it (DeviceGuard)
    hipStream_t stream = nullptr;
    std::map<std::string, DevBuf> bufs;
    std::shared_ptr<Arena> arena;  // set when a batch moved the buffers into its arena (they are not freed one by one then)
    // The plan's ~75 buffers are carved out of a few slabs (one hipMalloc / hipFree per 32 MB instead of one per buffer: a job list
    // whose array radii change from chunk to chunk creates and releases its plans inside the call, and 1200 hipFree calls per chunk
    // of 16 plans were 0.28 s of its 0.39 s).  A buffer that grows takes a new region; the slabs go when the plan goes, or when a
    // batch has moved every buffer into its arena.
    struct Slab { char* base; size_t size, used; };
    std::vector<Slab> slabs;
    bool slab_zeroed = false;      // the newest slab was zero-filled when it was taken
    static constexpr size_t SLAB_BYTES = (size_t)32 << 20;
    void* slab_take(size_t bytes) {
        bytes = (bytes + 255) / 256 * 256;
        if (slabs.empty() || slabs.back().used + bytes > slabs.back().size) {
            Slab sl{nullptr, (std::max(bytes, SLAB_BYTES) + SLAB_BYTES - 1) / SLAB_BYTES * SLAB_BYTES, 0};
            sl.base = static_cast<char*>(BlockPool::get().take(sl.size, &sl.size));
            // (one fill per slab instead of one per buffer: 66 hipMemsetAsync calls were 1.9 ms of a plan's set-up)
            static const bool fill = [] { const char* e = getenv("EMAGLS_SLAB_FILL"); return !(e && e[0] == '0'); }();
            if (!fill || hipMemsetAsync(sl.base, 0, sl.size, stream) != hipSuccess) { (void)hipGetLastError(); slab_zeroed = false; } else slab_zeroed = true;
            slabs.push_back(sl);
        }
        void* p = slabs.back().base + slabs.back().used;
        slabs.back().used += bytes;
        return p;
    }
    void release_slabs() {
        for (auto& sl : slabs) BlockPool::get().give(sl.base, sl.size);
        slabs.clear();
    }
    int64_t total_bytes = 0;
    // derived constants
    bool cplx_basis = false;      // element type of the internal SH machinery
    bool req_cplx = false;        // shDefinition == 'complex' was requested
    bool real_internal = false;   // complex request served by the real-arithmetic pipeline + a unitary channel transform
    int nfft = 0, P = 0, k_cut = 0, kcut0 = 0;
    int simOrder = 0, S = 0, C = 0, ldS = 0, nOut = 0;
    int simOrderOwn = 0;          // the design's own simulation order (getSMAIRMatrix.m:95); simOrder may be padded above it
    int64_t D = 0, ldD = 0, Dpad = 0, Dm = 0;  // Dm: matched direction count (FROM_ATF)
    bool hrir_smaller = true;
    bool out_cplx = false;
    int64_t out_rows = 0, out_cols = 0;
    int nWG = 0, nWG_dense = 0;   // workgroups of the launch-per-bin sweeps (MagLS / FromAtf: nWG; array designs: nWG_dense)
    // Gram route of the per-bin factorisation for the well-conditioned swept bins (factor.hip); switched off for good
    // when a run reports that the kr-based conditioning estimate was too optimistic (the plan is then re-executed)
    bool gram_route = true;
    // Routes of the per-bin factorisation (plan_routes): bins [1, hh_end) take the orthonormal S-space route (Householder QR +
    // Jacobi SVD) on the orders 0..n_h whose modal strength is above 1e-20 of the strongest there (S_h = (n_h+1)^2 rows); bins
    // [gram_from, P) take the Gram route (gramroute.hip) on all orders.  gram_floor: lower bound of gram_from that a device-side
    // conditioning check imposed (recovery).  g0: first bin whose direction-space operand G_k exists.
    int gram_from = 0, gram_floor = 0, hh_end = 0, n_h = 0, S_h = 0, ldS_h = 0, g0 = 0, nb_gram = 0;
    int nh_floor = 0;   // least number of orders on the Householder route (a lane batch gives all its designs the same routes)
    bool persist_suspended = false;   // sweep_persist switched off for ONE re-run (status word 4), restored afterwards
    bool sweep_persist = true;  // (EMAGLS_SWEEP_PERSIST=0 disables) one resident launch for all swept bins (sweep_persist.hip)
    // Operand synthesis (sweep_synth.hip): the resident sweep evaluates the slab of pwGrid_k.' of every bin itself from the angles
    // between HRIR directions and microphones instead of reading a materialised G_k (540 MB per design at config 3).  synth_want:
    // the design qualifies (built-in real SH machinery, <= 32 microphones, no covariance constraint); synth: it is in effect
    // (persistent sweep, no swept bin on the Householder route) -- plan_update_synth
    bool synth_want = false, synth = false;
    int synth_units = 0;          // antipodal microphone pairs + single microphones (set with the microphone grid)
    // the register-resident form of the synthesising sweep (sweep_reg.hip) took the last sweep of this plan (decided per launch:
    // reg_sweep_wanted); its argument block lies in device memory ("sweep_args"; the host copy tells when it has to be stored again)
    bool reg_sweep = false;
    std::vector<char> sweep_args_last;
    bool synth_block = false;     // a batch whose designs do not all qualify keeps every one of them on the materialised operands
    const emagls_plan* geo_from = nullptr;   // set while a geometry-sharing batch runs this plan's stages on plan 0's geometry
    emagls_batch* owner = nullptr;  // the batch this plan currently belongs to (cleared by either destructor)
    bool have_hrir_grid = false, have_mic_grid = false, have_hrirs = false, have_atfs = false, have_basis = false;
    uint64_t atf_side_version = 0;   // bumped when the grids or the ATF set are replaced (a FromAtf batch re-checks that its plans agree)
    bool diffuse = false;         // diffuseness (covariance) constraint after the sweep (render.hip: diffuse_constraint_kernel)
    bool custom_basis = false;    // the SH matrices come from the caller (a custom shFunction evaluated on the MATLAB side)
    bool wide = false;            // LS / MagLS with 33..64 channels (SH orders 5..7): the plain path of wide.hip
    // profiling
    int prof_level = 0;
    std::vector<std::string> stage_names;
    std::vector<hipEvent_t> stage_events;
    std::vector<double> stage_ms;
    std::vector<hipEvent_t> sweep_events;
    int sweep_launches = 0;
    bool executed = false;
    // hipGraph replay of the whole design (launch-bound: ~520 small kernels per execute)
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int eager_runs = 0;
    bool use_graph = true;
    hipGraph_t pre_graph = nullptr;          // batches: stages before the sweep, captured on the plan's own stream
    hipGraphExec_t pre_exec = nullptr;
    int nstreams = 1;
    int stage_order = 0;          // order of the stages before the sweep (emagls_pre_sweep): 0 branches, 1 / 2 the complementary single-stream orders of lane groups
    int pre_phase = 0;            // emagls_pre_sweep: 0 everything, 1 only what the sweep needs, 2 the rest (plan_defers_hh_route)
    // HRIR sets on ONE geometry through a plan of the 33..64-channel path (emagls_design_hrir_sets): what depends on the grids and the array only
    // -- G_k, the per-bin factors, Y_reg_inv_k: 19 of the 31 ms of a 64-capsule design -- is kept from the last clean run on the same grids
    bool geo_keep = false;                   // the caller runs sets of one geometry through this plan
    bool geo_skip = false;                   // (this execute: the geometry stages are skipped)
    uint64_t geo_done_version = ~0ull;       // atf_side_version of the last run whose flags came back clean
    uint64_t geo_run_version = ~0ull;        // ... of the last full run (promoted by plan_check_flags)
    bool defer_hh = false;        // plan_execute: what the captured stages before the sweep were captured with
    bool alone = false;           // the plan of a one-shot call (the device to itself, like a plan with forked stages)
    hipStream_t hh_stream = nullptr;   // the stream of the stages that run next to the sweep
    hipStream_t sync_stream = nullptr;  // stream whose completion means this plan's results are ready
    // fork/join inside one design: independent branches run on side streams (captured into the same graph)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};   // taken from the pool when a multi-stream execute first needs them (need_sides)
    bool owns_stream = true;      // false: `stream` belongs to the job slot that created the plan (one stream for all plans of a chunk)
    void need_sides(int n) { for (int i = 0; i < n - 1 && i < 3; ++i) if (!side[i]) side[i] = StreamPool::get().take(); }
    std::vector<hipEvent_t> sync_events;
    size_t sync_used = 0;
    hipEvent_t next_sync_event() {
        if (sync_used == sync_events.size()) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            sync_events.push_back(e);
        }
        return sync_events[sync_used++];
    }
    // make `waiter` wait for everything enqueued so far on `signaller`
    void depend(hipStream_t waiter, hipStream_t signaller) {
        if (waiter == signaller) return;  // same stream: already ordered
        hipEvent_t e = next_sync_event();
        HIP_CHECK(hipEventRecord(e, signaller));
        HIP_CHECK(hipStreamWaitEvent(waiter, e, 0));
    }

    ~emagls_plan() {
        if (owner) emagls_batch_forget(owner, this);
        // the slabs go back to a pool that other threads take from at once (no hipFree that would wait for pending work): nothing
        // enqueued by this plan may still be running on them
        if (stream) hipStreamSynchronize(stream);
        for (auto st : side) if (st) hipStreamSynchronize(st);
        if (hh_stream) hipStreamSynchronize(hh_stream);
        (void)hipGetLastError();
        for (auto& kv : bufs) if (kv.second.p && kv.second.owned) hipFree(kv.second.p);
        release_slabs();
        for (auto e : stage_events) hipEventDestroy(e);
        for (auto e : sweep_events) hipEventDestroy(e);
        for (auto e : sync_events) hipEventDestroy(e);
        for (auto st : side) StreamPool::get().give(st);
        StreamPool::get().give(hh_stream);
        if (graph_exec) hipGraphExecDestroy(graph_exec);
        if (graph) hipGraphDestroy(graph);
        if (pre_exec) hipGraphExecDestroy(pre_exec);
        if (pre_graph) hipGraphDestroy(pre_graph);
        if (owns_stream) StreamPool::get().give(stream);
    }
    void* alloc(const std::string& name, size_t bytes, bool zero = true) {
        if (bytes == 0) bytes = 16;
        auto it = bufs.find(name);
        if (it != bufs.end()) {   // re-allocation (a design's routes changed): keep what is large enough
            if (it->second.bytes >= bytes) {   // (the recorded size follows the request: lane batches compare and copy by it)
                total_bytes -= (int64_t)(it->second.bytes - bytes);
                it->second.bytes = bytes;
                return it->second.p;
            }
            if (it->second.owned) HIP_CHECK(hipFree(it->second.p));
            total_bytes -= (int64_t)it->second.bytes;
        }
        DevBuf b;
        b.p = slab_take((bytes + 15) / 16 * 16);  // (launch_zero works on whole 8-byte words)
        b.bytes = bytes;
        b.owned = false;   // (part of a slab, zero-filled when the slab was taken)
        if (zero && !slab_zeroed) HIP_CHECK(hipMemsetAsync(b.p, 0, bytes, stream));
        bufs[name] = b;
        total_bytes += (int64_t)bytes;
        return b.p;
    }
    template <typename T = void> T* get(const std::string& name) {
        auto it = bufs.find(name);
        if (it == bufs.end()) throw Error(EMAGLS_ERR_ARG, "internal: unknown buffer " + name);
        return reinterpret_cast<T*>(it->second.p);
    }
    bool has(const std::string& name) const { return bufs.count(name) != 0; }
    void upload(const std::string& name, const void* src, size_t bytes) {
        auto it = bufs.find(name);
        if (it == bufs.end() || it->second.bytes < bytes) throw Error(EMAGLS_ERR_ARG, "internal: upload size mismatch for " + name);
        HIP_CHECK(hipMemcpyAsync(it->second.p, src, bytes, hipMemcpyDefault, stream));
    }
    void mark(const char* name) {
        if (prof_level < 1) return;
        const size_t i = stage_names.size();
        stage_names.push_back(name);
        if (stage_events.size() <= i) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreate(&e));
            stage_events.push_back(e);
        }
        HIP_CHECK(hipEventRecord(stage_events[i], stream));
    }
};

struct emagls_batch {
    std::vector<emagls_plan*> plans;
    int device = -1;              // device of its plans
    // lanes: all plans have the same shape and their buffers sit `stride` bytes apart in one arena, so every
    // launch of the design pipeline covers the whole batch (grid.z = design)
    bool lanes = false;
    size_t stride = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;                    // false once the caller supplied the stream (emagls_batch_set_stream)
    // FromAtf subjects: the ATF side (spectra of the matched ATFs, per-bin factors) is computed by plan 0 and read by all plans when
    // they hold the same grids and ATF set (checked on the device whenever one of them was replaced)
    bool atf = false, atf_share = false, atf_inputs_same = false;
    uint64_t atf_checked_version = ~0ull;
    // array designs that differ only in their HRIR sets (same grids, array, orders): the geometry stages run once (opt-in,
    // emagls_batch_set_geometry_sharing; checked on the device whenever a grid was replaced)
    bool magls = false;     // MagLS / MagLS-2D plans (HRIR sets on one or several grids): batch_execute_magls
    bool geo_want = false, geo_share = false, geo_inputs_same = false;
    uint64_t geo_checked_version = ~0ull;
    int* cmp_flag = nullptr;
    int nstreams = 1;                          // lane mode: streams the stages before the sweep fork onto (emagls_batch_set_streams)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int prof_level = 0;
    hipEvent_t sweep_ev[2] = {nullptr, nullptr};
    hipGraph_t post_graph = nullptr;           // lane mode: the stages after the sweep (the sweep is launched directly)
    hipGraphExec_t post_exec = nullptr;
    bool side0_external = false;               // side[0] belongs to the caller (emagls_batch_set_side_stream)
    int order_hint = 0;                        // single-group batches: 1 / 2 = stage order of emagls_pre_sweep the caller asks for (emagls_batch_set_stage_order)
    int groups = 1;                            // lane groups before the sweep (ceil(designs / 8), up to 4: batch_execute_lanes)
    hipGraph_t graph2 = nullptr;               // the second lane group's stages before the sweep (on side[0])
    hipGraphExec_t graph2_exec = nullptr;
    hipGraph_t graphx[2] = {nullptr, nullptr};             // the third and fourth groups' (on side[1], side[2])
    hipGraphExec_t graphx_exec[2] = {nullptr, nullptr};
    hipGraph_t graph_hh[4] = {nullptr, nullptr, nullptr, nullptr};   // per lane group: the stages the sweep does not need (plan_defers_hh_route), next to the sweep
    hipGraphExec_t graph_hh_exec[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t hh_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    bool defer_hh = false;                     // what the captured graphs were captured with (batch_execute_lanes)
    bool alone = true;                         // the batch has the device to itself: the default of emagls_batch_create; the job scheduler clears it for the chunks of a list that keeps several in flight
    int eager_runs = 0;
    bool use_graph = true;
    void* sweep_args_dev = nullptr;            // argument blocks of the register-resident sweep, one per plan (sweep_reg.hip)
    std::vector<char> sweep_args_last;
    std::vector<hipEvent_t> events;
    size_t used = 0;
    hipEvent_t next_event() {
        if (used == events.size()) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            events.push_back(e);
        }
        return events[used++];
    }
    void depend(hipStream_t waiter, hipStream_t signaller) {
        hipEvent_t e = next_event();
        HIP_CHECK(hipEventRecord(e, signaller));
        HIP_CHECK(hipStreamWaitEvent(waiter, e, 0));
    }
    ~emagls_batch() {
        // a batch may still be in flight on a caller-owned stream (emagls_batch_set_stream): its graph execs and events must
        // outlive it (pool-owned streams are synchronised again when they are handed back)
        if (stream) hipStreamSynchronize(stream);
        for (auto e : events) hipEventDestroy(e);
        if (graph_exec) hipGraphExecDestroy(graph_exec);
        if (graph) hipGraphDestroy(graph);
        if (post_exec) hipGraphExecDestroy(post_exec);
        if (post_graph) hipGraphDestroy(post_graph);
        if (graph2_exec) hipGraphExecDestroy(graph2_exec);
        if (graph2) hipGraphDestroy(graph2);
        for (int i = 0; i < 2; ++i) { if (graphx_exec[i]) hipGraphExecDestroy(graphx_exec[i]); if (graphx[i]) hipGraphDestroy(graphx[i]); }
        for (int i = 0; i < 4; ++i) {
            if (hh_stream[i]) { hipStreamSynchronize(hh_stream[i]); emagls::pool_stream_give(hh_stream[i]); }
            if (graph_hh_exec[i]) hipGraphExecDestroy(graph_hh_exec[i]);
            if (graph_hh[i]) hipGraphDestroy(graph_hh[i]);
        }
        for (auto e : sweep_ev) if (e) hipEventDestroy(e);
        if (stream && own_stream) emagls::pool_stream_give(stream);
        for (int i = 0; i < 3; ++i) if (side[i] && !(i == 0 && side0_external)) { hipStreamSynchronize(side[i]); emagls::pool_stream_give(side[i]); }
        if (cmp_flag) hipFree(cmp_flag);
        if (sweep_args_dev) hipFree(sweep_args_dev);
        for (auto* p : plans) if (p) { p->sync_stream = nullptr; p->owner = nullptr; }
    }
};
// a plan of the batch is being destroyed before the batch: the batch must not touch it again
void emagls_batch_forget(emagls_batch* b, emagls_plan* p) {
    for (auto& q : b->plans) if (q == p) q = nullptr;
}

namespace {


// compute units of the current device (cached per device id)
int device_cu_count() {
    static std::mutex mu;
    static std::map<int, int> cache;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(dev);
    if (it != cache.end()) return it->second;
    int n = 0;
    HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cache[dev] = n;
    return n;
}

// (power-of-two FFT lengths run on the LDS FFTs of fft.hip, any other even length on its direct-DFT kernels)
void check_nfft(int nfft) {
    if (nfft < 8) throw Error(EMAGLS_ERR_UNSUPPORTED, "filter length below 4 is not supported");
}

// ---------------------------------------------------------------------------------------------
// plan construction: derive constants, allocate every device buffer once
// ---------------------------------------------------------------------------------------------
// kinds that run the array-model pipeline (simulated array -> per-bin factor -> sweep)
static inline bool magls_kind(int k) { return k == EMAGLS_KIND_MAGLS || k == EMAGLS_KIND_MAGLS_2D; }
static inline bool array_kind(int k) { return k == EMAGLS_KIND_EMAGLS || k == EMAGLS_KIND_EMAGLS2 || k == EMAGLS_KIND_EMA_CH || k == EMAGLS_KIND_EMA_SH; }
// evaluation points of the SH rotation fit (emash.hip): enough to resolve order N exactly
static inline int ema_sh_npts(int C) { return 4 * C + 8; }

// Smallest order n such that every order above it contributes less than 1e-20 of the strongest mode to pwGrid at kr = x:
// |b_n(x)| (2n+1) / |b_0| <= x^n / (2n-1)!! (2n+1) for the rigid sphere (|j_n(x)| <= x^n / (2n+1)!!; the Wronskian form of b_n
// divides by x^2 |h_n'(x)| >= (n+1) (2n-1)!! / x^n).  Dropping those orders perturbs the bin's matrix by 1/200 of its own
// FP64 rounding error: the reference's LAPACK SVD cannot tell the difference.
constexpr double ORDER_NOISE = 1e-18;
int orders_above_noise(double x, int nmax) {
    double term = 1.0;   // x^n / (2n-1)!!
    for (int n = 1; n <= nmax; ++n) {
        term *= x / (double)(2 * n - 1);
        if ((double)n > x && term * (2 * n + 1) < ORDER_NOISE) return n - 1;
    }
    return nmax;
}

int emagls_gram_from(const emagls_plan& p);
// routes of the per-bin factorisation (see emagls_plan): derived from kr only, so that every rank / replay takes the same
void plan_routes(emagls_plan& p) {
    const emagls_design_desc& d = p.d;
    const int k0 = std::max(p.kcut0, 1);
    p.gram_from = emagls_gram_from(p);
    // EMAinSH has no radial terms in its model: pwGrid_k is well conditioned at every bin (emash.hip) and all bins take the Gram route
    if (d.kind == EMAGLS_KIND_EMA_SH) p.gram_from = 1;
    else if (p.gram_from > 0 && p.gram_from < p.gram_floor) p.gram_from = p.gram_floor < p.P ? p.gram_floor : 0;
    p.hh_end = p.gram_from > 0 ? p.gram_from : p.P;
    const double f_h = (double)(p.hh_end - 1) * (d.fs / 2.0) / (double)(p.P - 1);
    int n_min = 0;   // the S-space factor needs at least as many rows as channels
    while ((n_min + 1) * (n_min + 1) < p.C) ++n_min;
    p.n_h = std::min(p.simOrder, std::max({orders_above_noise(2.0 * kPi * f_h / C_SOUND * d.mic_radius, p.simOrder), n_min, p.nh_floor}));
    p.S_h = (p.n_h + 1) * (p.n_h + 1);
    p.ldS_h = round_up(p.S_h, 64);
    if (p.S_h > 768)
        throw Error(EMAGLS_ERR_UNSUPPORTED, "the ill-conditioned low bins of this design need more than 27 orders on the orthonormal route "
                                            "(Gram route off or moved up by a conditioning check): not supported in this build");
    if (d.kind == EMAGLS_KIND_EMA_SH) { p.hh_end = 1; p.n_h = n_min; p.S_h = (n_min + 1) * (n_min + 1); p.ldS_h = round_up(p.S_h, 64); }
    // the orthonormal route factors the first S_h columns of the grid's SH matrix (Cholesky of their Gram block): they must be
    // independent.  The columns beyond S_h only enter through products (Gram matrix, order terms), so D < S is no obstacle.
    if (p.D < p.S_h)
        throw Error(EMAGLS_ERR_UNSUPPORTED, "fewer HRIR directions than the SH channels of the orthonormal route (the low bins need "
                                            "(n_h + 1)^2 independent columns of the grid's SH matrix)");
    p.g0 = (p.gram_from > 0 && p.gram_from < k0) ? p.gram_from : k0;
    if (p.diffuse) p.g0 = 1;   // the constraint renders the HRTFs of every solved bin: G_k from the first one
    p.nb_gram = p.gram_from > 0 ? p.P - p.gram_from : 0;
}
// Row order of the microphones in the synthesising sweep: smap[0..M) = microphone of row r, smap[32] = antipodal pairs (rows 2u,
// 2u + 1), smap[33] = single microphones after them.  Two microphones are a pair when their unit vectors cancel to a few ulps:
// cos(d, j') = -cos(d, j) then holds to the rounding error of either cosine, and one polynomial evaluation serves both.
// EMAGLS_SYNTH_PAIRS=0: no pairing.
static void synth_pairing(const double* azi, const double* zen, int M, int* smap) {
    static const bool pairs_on = [] { const char* e = getenv("EMAGLS_SYNTH_PAIRS"); return !(e && e[0] == '0'); }();
    std::vector<double> u((size_t)3 * M);
    for (int j = 0; j < M; ++j) {
        u[3 * j] = std::sin(zen[j]) * std::cos(azi[j]); u[3 * j + 1] = std::sin(zen[j]) * std::sin(azi[j]); u[3 * j + 2] = std::cos(zen[j]);
    }
    std::vector<int> partner((size_t)M, -1);
    const double tol = 8.0 * 2.220446049250313e-16;
    if (pairs_on && M <= 32)
        for (int j = 0; j < M; ++j) {
            if (partner[j] >= 0) continue;
            for (int k = j + 1; k < M; ++k) {
                if (partner[k] >= 0) continue;
                if (std::fabs(u[3 * j] + u[3 * k]) <= tol && std::fabs(u[3 * j + 1] + u[3 * k + 1]) <= tol && std::fabs(u[3 * j + 2] + u[3 * k + 2]) <= tol) {
                    partner[j] = k; partner[k] = j;
                    break;
                }
            }
        }
    for (int i = 0; i < 34; ++i) smap[i] = 0;
    int row = 0, npr = 0, nsg = 0;
    for (int j = 0; j < M && j < 32; ++j) if (partner[j] > j) { smap[row++] = j; smap[row++] = partner[j]; ++npr; }
    for (int j = 0; j < M && j < 32; ++j) if (partner[j] < 0) { smap[row++] = j; ++nsg; }
    smap[32] = npr; smap[33] = nsg;
}
// EMAGLS_SWEEP_SYNTH=0: every design on the materialised operands (dspace_g + sweep_persist_kernel)
// (read when a plan is created: a test switches forms inside one process)
static bool synth_enabled() { const char* e = getenv("EMAGLS_SWEEP_SYNTH"); return !(e && e[0] == '0'); }
static bool plan_persist_possible(const emagls_plan& p) {
    const int64_t Dh = (p.d.kind == EMAGLS_KIND_FROM_ATF) ? p.Dm : p.D;
    if (const char* e = getenv("EMAGLS_SWEEP_PERSIST")) if (e[0] == '0') return false;
    return !p.wide && persist_sweep_fits((int)Dh, p.C, 1);
}
// Is the synthesising sweep in effect?  Needs the persistent form (all workgroups resident) and every swept bin on the Gram
// route (the ill-conditioned swept bins of tiny arrays read Y_reg_inv_k from memory: they keep the materialised operands).
void plan_update_synth(emagls_plan& p) {
    const int k0 = std::max(p.kcut0, 1);
    p.synth = p.synth_want && !p.synth_block && p.sweep_persist && p.hh_end <= k0 && p.gram_from > 0 && k0 < p.P &&
              synth_sweep_fits((int)p.D, (int)p.d.nmics, p.simOrder + 1, 1);
}
// buffers whose size depends on the routes (re-entered when a conditioning check moves the routes: alloc keeps what is large enough)
void plan_alloc_routes(emagls_plan& p) {
    const bool cb = p.cplx_basis;
    const int k0 = std::max(p.kcut0, 1);
    const int ls_end = std::max(std::min(p.kcut0, p.P), 1);
    const int nOrd = p.simOrder + 1;
    p.alloc("R", esz(cb) * (size_t)p.S_h * p.S_h);                     // Cholesky factor of the leading block of Gy
    p.alloc("Rinv", esz(cb) * (size_t)ceil_div(p.S_h, 32) * 32 * 32);
    if (!cb) {   // complex copies of R and of its diagonal-block inverses (row solves of complex rows in the real basis)
        p.alloc("Rc", sizeof(cplx) * (size_t)p.S_h * p.S_h);
        p.alloc("Rinvc", sizeof(cplx) * (size_t)ceil_div(p.S_h, 32) * 32 * 32);
    }
    p.alloc("Tn", esz(cb) * (size_t)(p.n_h + 1) * p.C * p.ldS_h);
    p.alloc("Hq", sizeof(cplx) * (size_t)2 * ls_end * p.ldS_h);
    p.alloc("Hyp", sizeof(double) * hy_mfma_workspace_doubles(ls_end, p.S_h, cb));
    p.alloc("HcT", sizeof(double) * (size_t)hy_mfma_kpad((int)p.D) * round_up(4 * ls_end, 64));   // (rows >= D stay zero)
    p.alloc("Z", sizeof(cplx) * (size_t)p.hh_end * p.C * p.ldS_h);
    p.alloc("Vws", sizeof(cplx) * (size_t)p.hh_end * p.C * p.ldS_h);
    plan_update_synth(p);
    // (the synthesising sweep and its least-squares bins evaluate their operands themselves: no G_k in memory)
    const int g_end = p.synth ? p.g0 : p.P;
    p.alloc("G", sizeof(cplx) * ((size_t)std::max(g_end - p.g0, 1) * p.C + 32) * p.ldD, false);  // + 32 rows: the persistent sweep loads all 32 slab rows of a bin unconditionally
    if (p.synth_want) {
        const int M = (int)p.d.nmics;
        p.alloc("bsc", sizeof(cplx) * (size_t)p.P * synth_nord_pad(nOrd));
        p.alloc("Pm", sizeof(double) * 32 * 32);
        if (!p.has("smap")) {   // (identity order until the microphone grid arrives)
            int smap[34] = {0};
            for (int j = 0; j < 32; ++j) smap[j] = j < M ? j : 0;
            smap[33] = M;
            p.alloc("smap", sizeof smap);
            p.upload("smap", smap, sizeof smap);
            p.synth_units = M;
            HIP_CHECK(hipStreamSynchronize(p.stream));
        }
        p.alloc("Mt", sizeof(cplx) * ((size_t)p.P * M * M + 1024));
        p.alloc("Winit", sizeof(cplx) * 64);
        p.alloc("Usw", sizeof(cplx) * (size_t)synth_ls_chunks((int)p.D) * 2 * p.P * 32);   // [chunk][e][bin][32]: the chain's totals use chunk 0
    }
    p.alloc("Yri", sizeof(cplx) * (size_t)std::max(p.hh_end - k0, 1) * p.C * p.ldD, false);    // only Householder-route bins can be flagged ill-conditioned
    if (p.nb_gram > 0) {
        const int ldK = round_up(p.C * p.C, 64), Kp = round_up(nOrd * nOrd, 4);
        p.alloc("Fg", esz(cb) * (size_t)nOrd * p.C * p.ldS);
        p.alloc("Kmat", sizeof(double) * (size_t)Kp * ldK);                                   // (rows beyond nOrd^2 stay zero)
        p.alloc("Cf", sizeof(double) * (size_t)Kp * round_up(p.P, 64));                       // (sized for every bin: the routes may move)
        p.alloc("Apk", sizeof(double) * (size_t)p.P * ldK);
    }
}

thread_local hipStream_t g_plan_stream_shared = nullptr;   // set by the job scheduler around the creation of a chunk's plans
void plan_setup(emagls_plan& p) {
    const emagls_design_desc& d = p.d;
    if (d.kind < EMAGLS_KIND_LS || d.kind > EMAGLS_KIND_EMA_SH) throw Error(EMAGLS_ERR_ARG, "unknown design kind");
    if (d.basis != EMAGLS_BASIS_REAL && d.basis != EMAGLS_BASIS_COMPLEX) throw Error(EMAGLS_ERR_ARG, "shDefinition must be 'real' or 'complex'");
    if (d.ndirs < 1 || d.nsamp < 1) throw Error(EMAGLS_ERR_ARG, "empty HRIR set");
    if (d.kind != EMAGLS_KIND_FROM_ATF && d.order < 0) throw Error(EMAGLS_ERR_ARG, "negative SH order");
    HIP_CHECK(hipGetDevice(&p.device));
    const auto t_setup0 = std::chrono::steady_clock::now();
    // (the plans of a job chunk share the slot's stream -- hipStreamCreate was 3 ms of a plan's set-up, four streams each --; the side
    // streams of a multi-stream execute are taken when one first asks for them)
    if (g_plan_stream_shared) { p.stream = g_plan_stream_shared; p.owns_stream = false; }
    else p.stream = StreamPool::get().take();
    if (const char* ng = getenv("EMAGLS_NO_GRAPH")) p.use_graph = !(ng[0] == '1');
    if (const char* es = getenv("EMAGLS_EAGER_SIDES")) if (es[0] == '1') p.need_sides(4);   // (experiments: round 5's four streams per plan)
    const auto t_setup1 = std::chrono::steady_clock::now();
    if (const char* ns = getenv("EMAGLS_STREAMS")) p.nstreams = std::max(1, std::min(4, atoi(ns)));
    p.req_cplx = d.basis == EMAGLS_BASIS_COMPLEX;
    // Complex-basis eMagLS / eMagLS2 designs run in real arithmetic.  With Y_c = Y_r T (T unitary, block diagonal per order)
    // smair_c = T_N^H smair_r T and pwGrid_c = T_N^H pwGrid_r, hence Y_reg_inv_c = Y_reg_inv_r T_N, the angles
    // W(k-1,:) pwGrid are the same and W_c(k,:) = W_r(k,:) T_N for every solved bin (lib/getEMagLsFilters.m:87-103); the DC
    // rule and the SH conjugate rule (:109-118) act on W_c and stay in the epilogue.  eMagLS2 is basis free (T cancels).
    // The real pipeline has a 3x cheaper Gram and half the bytes in T_n and QT: 1460 vs 1295 sets/s at config 3.
    p.custom_basis = d.custom_basis != 0;
    p.diffuse = d.diffuseness != 0;
    if (p.diffuse && (d.kind == EMAGLS_KIND_LS || d.kind == EMAGLS_KIND_FROM_ATF))
        throw Error(EMAGLS_ERR_ARG, "the diffuseness constraint applies to MagLS, eMagLS, eMagLS2 and the EMA variant");
    if (p.custom_basis && (d.kind == EMAGLS_KIND_FROM_ATF || d.kind == EMAGLS_KIND_EMA_CH || d.kind == EMAGLS_KIND_MAGLS_2D))
        throw Error(EMAGLS_ERR_UNSUPPORTED, "caller-supplied SH matrices are available for LS, MagLS, eMagLS and eMagLS2 designs");
    // (a caller-supplied complex basis need not be ours rotated by T: it takes the complex-arithmetic pipeline)
    p.real_internal = p.req_cplx && !p.custom_basis && (d.kind == EMAGLS_KIND_EMAGLS || d.kind == EMAGLS_KIND_EMAGLS2);
    if (const char* e = getenv("EMAGLS_REAL_INTERNAL")) if (e[0] == '0') p.real_internal = false;
    p.cplx_basis = p.req_cplx && !p.real_internal;
    p.D = d.ndirs;
    p.ldD = round_up(p.D, 64);
    const bool cb = p.cplx_basis;

    p.alloc("hL", sizeof(double) * d.nsamp * d.ndirs, false);
    p.alloc("hR", sizeof(double) * d.nsamp * d.ndirs, false);
    p.alloc("hrir_azi", sizeof(double) * p.D, false);
    p.alloc("hrir_zen", sizeof(double) * p.D, false);
    p.alloc("flag", sizeof(int) * NFLAG);

    if (d.kind == EMAGLS_KIND_LS) p.alloc("grpd", sizeof(double) * 2);
    if (d.kind != EMAGLS_KIND_LS) {
        if (d.len < d.nsamp)
            throw Error(EMAGLS_ERR_ARG, magls_kind(d.kind) ? "HRIR len too short" : "len too short");
        if (!(d.fs > 0)) throw Error(EMAGLS_ERR_ARG, "fs must be positive");
        p.nfft = (int)std::min<int64_t>(NFFT_MAX_LEN, 2 * d.len);
        check_nfft(p.nfft);
        if (d.len % 2) throw Error(EMAGLS_ERR_ARG, "filter length must be even");
        // nfft is capped at NFFT_MAX_LEN: a longer filter makes the reference index wMlsL(n_shift-len/2+1 : n_shift+len/2) with a
        // non-positive start (lib/getEMagLsFilters.m:135-136) and fail; the kernels would read outside their LDS buffers.
        if (d.len > p.nfft)
            throw Error(EMAGLS_ERR_ARG, "len exceeds the oversampled FFT length min(2048, 2*len): the reference fails with an index error");
        p.P = p.nfft / 2 + 1;
        const double f2 = (d.fs / 2.0) / (double)(p.P - 1);  // f(2) of linspace(0, fs/2, P)
        const double f_cut = (d.kind == EMAGLS_KIND_FROM_ATF) ? d.f_trans : std::max(F_CUT_MIN_FREQ, 500.0 * d.order);
        p.k_cut = (int)std::ceil(f_cut / f2);
        if (p.k_cut < 1) p.k_cut = 1;
        p.kcut0 = std::min(p.k_cut - 1, p.P);  // 0-based index of the first magnitude-least-squares bin
        if (magls_kind(d.kind) && p.kcut0 < 1) throw Error(EMAGLS_ERR_ARG, "k_cut must be at least 2");
        p.alloc("tw", sizeof(cplx) * p.nfft);
        p.alloc("grpd", sizeof(double) * (2 + 4 * (size_t)p.P));   // the two delays, then the delay phases [2][P] (grpdelay_median_kernel)
        if (p.diffuse) p.alloc("Hfull", sizeof(cplx) * (size_t)2 * p.P * p.ldD);   // time-aligned complex HRTFs of every bin
        p.alloc("dirsum", sizeof(double) * 2 * d.nsamp * hrir_dirsum_chunks(d.ndirs));
    }

    const int N = d.order;
    if (d.kind == EMAGLS_KIND_LS || magls_kind(d.kind)) {
        p.simOrder = N;
        // getMagLsFilters2D.m:49: Y_conj = getCH(order, azi)' has 2*order+1 rows (the numHarmonics of :47 is never used)
        p.S = d.kind == EMAGLS_KIND_MAGLS_2D ? 2 * N + 1 : (N + 1) * (N + 1);
        p.C = p.S;
        p.nOut = p.S;
        // up to 32 channels: the tuned kernels (register tiles, the persistent sweep); 33..256 (SH orders 5..15, CH orders 16..127): the
        // plain path of wide.hip -- pinv(Y_conj) from the inverse of the Gram matrix, one sweep launch per bin (above 64 channels its
        // loop forms)
        p.wide = p.S > 32;
        if (p.S > 256) throw Error(EMAGLS_ERR_UNSUPPORTED, d.kind == EMAGLS_KIND_MAGLS_2D ? "CH order above 127 is not supported in this build"
                                                                                            : "SH order above 15 is not supported for LS/MagLS in this build");
        if (p.S > 64 && p.diffuse) throw Error(EMAGLS_ERR_UNSUPPORTED, "more than 64 channels: no covariance constraint in this build");
        if (p.D < p.S) throw Error(EMAGLS_ERR_UNSUPPORTED, "fewer HRIR directions than SH channels");
    } else if (array_kind(d.kind)) {
        if (!(d.mic_radius > 0) || d.nmics < 1) throw Error(EMAGLS_ERR_ARG, "invalid array geometry");
        // getSMAIRMatrix.m:95: max(params.order, ceil(fs*pi*r/C)).  lib/getEMagLs2Filters.m:51-63 leaves params.order unset, so
        // getSMAIRMatrix.m:39-41 defaults it to 4 there: for eMagLS2 `order` only sets f_cut (:47), never the simulation order.
        const int smair_order = d.kind == EMAGLS_KIND_EMAGLS2 ? SMAIR_DEFAULT_ORDER : N;
        p.simOrderOwn = std::max(smair_order, (int)std::ceil(d.fs * kPi * d.mic_radius / C_SOUND));
        // sim_order_pad: simulate on more orders than the design's own, with b_n = 0 above its own order -- the same sum, so
        // the same filters; array radii of neighbouring simulation-order classes then have one shape and share a lane batch
        if (d.sim_order_pad < 0) throw Error(EMAGLS_ERR_ARG, "negative sim_order_pad");
        if (d.sim_order_pad > 0 && (p.custom_basis || d.kind == EMAGLS_KIND_EMA_SH))
            throw Error(EMAGLS_ERR_UNSUPPORTED, "sim_order_pad is available for eMagLS / eMagLS2 / EMAinCH designs on the built-in SH basis");
        p.simOrder = std::max(p.simOrderOwn, d.sim_order_pad);
        p.S = (p.simOrder + 1) * (p.simOrder + 1);
        p.nOut = d.kind == EMAGLS_KIND_EMA_CH ? 2 * N + 1 : (N + 1) * (N + 1);   // EMAinCH.m:66: numHarmonics = 2*order+1
        if (d.kind == EMAGLS_KIND_EMA_SH && d.nmics < 2 * N + 1)
            throw Error(EMAGLS_ERR_UNSUPPORTED, "fewer microphones than circular harmonics (2*order+1)");
        p.C = d.kind == EMAGLS_KIND_EMAGLS2 ? (int)d.nmics : p.nOut;
        // up to 32 channels / microphones: the tuned per-bin kernels.  33..64 (a 64-capsule array; SH orders 5..7 in the SH domain):
        // the plain S-space path of wide_array.hip -- real-arithmetic pipeline, one design at a time (any simulation order the
        // narrow path takes: 64 microphones at 7 / 8 / 10 cm agree with the oracle to 1e-10, tools/experiments/wide_radius.py)
        if (p.C > 32) {
            if (p.C > 64) throw Error(EMAGLS_ERR_UNSUPPORTED, "more than 64 output channels is not supported in this build");
            if (d.kind != EMAGLS_KIND_EMAGLS && d.kind != EMAGLS_KIND_EMAGLS2 && d.kind != EMAGLS_KIND_EMA_SH)
                throw Error(EMAGLS_ERR_UNSUPPORTED, "more than 32 output channels: eMagLS / eMagLS2 / EMAinSH only");
            if (p.custom_basis || d.sim_order_pad > 0)
                throw Error(EMAGLS_ERR_UNSUPPORTED, "more than 32 output channels: built-in SH basis, no padding");
            // (EMAinSH orders 5..7 factor the direction-space operands themselves -- execute_ema_sh_wide -- in either basis)
            if (d.kind != EMAGLS_KIND_EMA_SH && p.req_cplx && !p.real_internal)
                throw Error(EMAGLS_ERR_UNSUPPORTED, "more than 32 output channels: the real-arithmetic pipeline only");
            if (d.kind == EMAGLS_KIND_EMA_SH && p.diffuse) throw Error(EMAGLS_ERR_UNSUPPORTED, "EMAinSH above order 4: no covariance constraint in this build");
            p.wide = true;
        }
        if (p.simOrder > 85) throw Error(EMAGLS_ERR_UNSUPPORTED, "simulation order above 85 (array radius > ~19.3 cm at 48 kHz) is not supported: the reference's own getSH overflows there (factorials beyond 170!)");
        // (fewer directions than simulated SH channels are fine as long as the orders of the orthonormal route are covered:
        // plan_routes checks D >= S_h.  The wide path orthogonalises all S columns.)
        if (p.D < p.S && (p.wide || d.kind == EMAGLS_KIND_EMA_SH)) throw Error(EMAGLS_ERR_UNSUPPORTED, "fewer HRIR directions than simulated SH channels");
        // (rank-deficient array model on the 33..64-channel path, e.g. 49 microphones on a 2 cm sphere at 16 kHz -- 25 simulated SH channels: the
        // reference's clipped inverse is then 100 / s_max times singular vectors of rounding noise; launch_wa_factor)
        if (p.wide && d.kind != EMAGLS_KIND_EMA_SH && p.S < p.C)
            throw Error(EMAGLS_ERR_UNSUPPORTED, "33..64 channels with fewer simulated SH channels than channels (rank-deficient array model: the reference's clipped inverse is rounding noise there) is not supported");
        if (p.D < p.C) throw Error(EMAGLS_ERR_UNSUPPORTED, "fewer HRIR directions than channels");
        if (d.kind != EMAGLS_KIND_EMAGLS2 && d.kind != EMAGLS_KIND_EMA_SH && d.nmics < p.nOut)
            throw Error(EMAGLS_ERR_UNSUPPORTED, "fewer microphones than output channels");
    } else {
        if (d.nmics < 1 || d.natf < 1 || d.atf_taps < 1) throw Error(EMAGLS_ERR_ARG, "invalid ATF set");
        p.C = (int)d.nmics;
        // up to 32 microphones on the Gram route (the M x M factors of the persistent sweep's form); the dense route behind its
        // conditioning flag -- QR + Jacobi of the Dm x M matrix itself -- holds up to 8 columns at this row count (factor.hip)
        // (33..64 microphones: the plain per-bin path of wide_array.hip on the matched ATF matrices themselves, one subject at a time)
        if (p.C > 64) throw Error(EMAGLS_ERR_UNSUPPORTED, "more than 64 ATF microphones is not supported in this build");
        p.wide = p.C > 32;
        p.hrir_smaller = d.ndirs <= d.natf;  // min([a b]) returns the first index on ties (FromAtf.m:62)
        p.Dm = p.hrir_smaller ? d.ndirs : d.natf;
        // (up to 3072 matched directions: resident sweep; above: the Gram route with one launch per bin, sweep_half_kernel walking
        // several slabs per workgroup; the dense route -- QR of the Dm x M matrix itself -- holds 4096 rows)
        if (p.Dm > 65536) throw Error(EMAGLS_ERR_UNSUPPORTED, "more than 65536 matched directions is not supported in this build");
        if (p.Dm < p.C) throw Error(EMAGLS_ERR_UNSUPPORTED, "fewer directions than microphones");
    }
    p.ldS = round_up(std::max(p.S, 1), 64);

    if (d.kind != EMAGLS_KIND_FROM_ATF) {
        // ---- SH machinery on the HRIR grid
        p.Dpad = std::max<int64_t>(gram_dpad(p.D, p.S), hy_mfma_kpad((int)p.D));   // (rows D..Dpad of Yc are zero)
        p.alloc("sh_tab", sizeof(double) * sh_coeff_count(p.simOrder));
        p.alloc("Ycm", esz(cb) * (size_t)p.S * p.ldD);                 // [S][ldD] column-major SH matrix
        p.alloc("Yc", esz(cb) * (size_t)p.Dpad * p.ldS);               // [Dpad][ldS] conj(Y), direction-major
        p.alloc("Gp", esz(cb) * (size_t)gram_ksplit(p.D, p.S) * p.S * p.S);
        if (!array_kind(d.kind)) {   // (array designs: R covers the Householder-route orders only, plan_alloc_routes; Q is never formed)
            p.alloc("R", esz(cb) * (size_t)p.S * p.S);
            p.alloc("Q", esz(cb) * (size_t)p.D * p.ldS);
            p.alloc("Rinv", esz(cb) * (size_t)ceil_div(p.S, 32) * 32 * 32);
        }
    }
    if (d.kind == EMAGLS_KIND_LS || magls_kind(d.kind)) {
        p.alloc("Rb", sizeof(cplx) * (size_t)p.C * p.ldS);             // R as [c][s] complex
        p.alloc("Zb", sizeof(cplx) * (size_t)p.C * p.ldS);
        p.alloc("Vws", sizeof(cplx) * (size_t)p.C * p.ldS);
        p.alloc("sv", sizeof(double) * p.C);
        p.alloc("tauw", sizeof(double) * p.C);
        p.alloc("R2w", sizeof(cplx) * (size_t)p.C * p.C);
        p.alloc("Nw", sizeof(cplx) * (size_t)p.C * p.C);
        p.alloc("Ypinv", esz(cb) * (size_t)p.C * p.ldD);
        if (p.wide) p.alloc("Mg", sizeof(cplx) * (size_t)p.S * p.S);   // (Y^T conj(Y))^-1
        if (p.S > 64) p.alloc("Mgw", sizeof(cplx) * (size_t)p.S * p.S + 2 * sizeof(double));   // R^-1 and the certificate's norms (wide.hip: 65..256 channels)
        if (d.kind == EMAGLS_KIND_LS) {
            p.out_rows = d.nsamp;
        } else {
            p.alloc("Xc", esz(cb) * (size_t)p.S * p.ldD);              // Y_conj as [c][d]
        }
    }
    if (array_kind(d.kind)) {
        const int M = (int)d.nmics;
        const int ldM = round_up(M, 64);
        p.alloc("mic_azi", sizeof(double) * M, false);
        p.alloc("mic_zen", sizeof(double) * M, false);
        p.alloc("Ymic_cm", esz(cb) * (size_t)p.S * M);                 // [S][M]
        p.alloc("Ymic_rm", esz(cb) * (size_t)ldM * p.ldS);             // [M][ldS]
        p.alloc("E", esz(cb) * (size_t)p.C * p.ldS);                   // [C][ldS]
        if (d.kind == EMAGLS_KIND_EMA_SH || (d.kind != EMAGLS_KIND_EMAGLS2 && p.nOut <= 32)) {   // (EMAinSH: pinv of the 2N+1 circular harmonics, any order)
            p.alloc("Ylo_c", sizeof(cplx) * (size_t)p.nOut * ldM);     // [nOut][ldM] complex copy of Y_Lo^T
            p.alloc("Zlo", sizeof(cplx) * (size_t)p.nOut * ldM);
            p.alloc("Vlo", sizeof(cplx) * (size_t)p.nOut * ldM);
            p.alloc("tau_lo", sizeof(double) * p.nOut);
            p.alloc("R2_lo", sizeof(cplx) * (size_t)p.nOut * p.nOut);
            p.alloc("N_lo", sizeof(cplx) * (size_t)p.nOut * p.nOut);
        }
        if (d.kind == EMAGLS_KIND_EMA_SH) {
            const int npts = ema_sh_npts(p.C), ldP = round_up(npts, 64);   // (C <= 64: orders up to 7, checked above)
            const int64_t ldA = round_up((int64_t)(p.D + 1) * npts, 64);
            p.alloc("Ech", esz(cb) * (size_t)(2 * N + 1) * p.ldS);          // pinv(CH(micAzi)) Y_mic
            p.alloc("sh_tab_lo", sizeof(double) * sh_coeff_count(N));      // recurrence table of the output order (its layout depends on the order)
            p.alloc("hrir_zen_eq", sizeof(double) * p.D, false);           // pi/2: the horizontal projection of the HRIR grid
            p.alloc("nnm_azi", sizeof(double) * p.C, false);
            p.alloc("nnm_zen", sizeof(double) * p.C, false);
            p.alloc("Ypts", esz(cb) * (size_t)p.C * p.C);
            p.alloc("rot_azi", sizeof(double) * (size_t)ldA, false);
            p.alloc("rot_zen", sizeof(double) * (size_t)ldA, false);
            p.alloc("Arot", esz(cb) * (size_t)p.C * ldA, false);           // SHs of order N at all rotated points (and the fixed set)
            p.alloc("Bc", sizeof(cplx) * (size_t)p.C * ldP);
            p.alloc("Zb", sizeof(cplx) * (size_t)p.C * ldP);
            p.alloc("Vb", sizeof(cplx) * (size_t)p.C * ldP);
            p.alloc("tau_b", sizeof(double) * p.C);
            p.alloc("R2_b", sizeof(cplx) * (size_t)p.C * p.C);
            p.alloc("N_b", sizeof(cplx) * (size_t)p.C * p.C);
            p.alloc("Rot", esz(cb) * (size_t)p.D * p.C * p.C, false);
            std::vector<double> eq((size_t)p.D, kPi / 2.0), na((size_t)p.C, 0.0), nz((size_t)p.C, kPi / 2.0);
            for (int c = 0; c < p.C; ++c) {   // one azimuth per channel at which its circular harmonic is 1 (or sqrt 2)
                int n = 0;
                while ((n + 1) * (n + 1) <= c) ++n;
                const int m = c - n * n - n;
                na[c] = (!cb && m < 0) ? kPi / (2.0 * -m) : 0.0;
            }
            p.upload("hrir_zen_eq", eq.data(), sizeof(double) * p.D);
            p.upload("nnm_azi", na.data(), sizeof(double) * p.C);
            p.upload("nnm_zen", nz.data(), sizeof(double) * p.C);
            HIP_CHECK(hipStreamSynchronize(p.stream));   // (the host vectors go out of scope)
        }
        p.alloc("kr", sizeof(double) * p.P, false);
        p.alloc("nvalid", sizeof(int) * 4);
        p.upload("nvalid", &p.simOrderOwn, sizeof(int));
        p.alloc("bn", sizeof(cplx) * (size_t)p.P * (p.simOrder + 1));
        if (p.wide && d.kind == EMAGLS_KIND_EMA_SH) {
            // EMAinSH orders 5..7 (36..64 channels): G_k of every bin materialised, factored in place of a common S-space
            // (wide_array.hip on the D x C operand itself, like FromAtf above 32 microphones), one sweep launch per bin
            const size_t nb = (size_t)p.P - 1, nOrdW = (size_t)p.simOrder + 1;
            p.alloc("QT", esz(cb) * nOrdW * p.C * p.ldD);
            p.alloc("G", sizeof(cplx) * (nb * p.C + 32) * p.ldD, false);
            p.alloc("Yri", sizeof(cplx) * (nb * p.C + 32) * p.ldD, false);
            p.alloc("Bw", sizeof(cplx) * nb * p.C * p.ldD, false);
            p.alloc("Vw", sizeof(cplx) * nb * p.C * p.ldD, false);
            p.alloc("tauw", sizeof(double) * nb * p.C);
            p.alloc("R2w", sizeof(cplx) * nb * p.C * p.C);
            p.alloc("Nw", sizeof(cplx) * nb * p.C * p.C);
            p.alloc("sv", sizeof(double) * (size_t)p.P * p.C);
            p.alloc("jsweeps", sizeof(int) * (size_t)p.P);
            p.g0 = 1;
        } else if (p.wide) {   // wide_array.hip: every bin on the S-space route in global memory, Y_reg_inv of every bin materialised
            const size_t nb = (size_t)p.P - 1, nOrdW = (size_t)p.simOrder + 1;
            p.alloc("R", sizeof(double) * (size_t)p.S * p.S);
            p.alloc("Rinv", sizeof(double) * (size_t)ceil_div(p.S, 32) * 32 * 32);
            p.alloc("Q", sizeof(double) * (size_t)p.D * p.ldS);
            p.alloc("Tn", sizeof(double) * nOrdW * p.C * p.ldS);
            p.alloc("QT", sizeof(double) * nOrdW * p.C * p.ldD);
            p.alloc("G", sizeof(cplx) * (nb * p.C + 32) * p.ldD, false);
            p.alloc("Yri", sizeof(cplx) * (nb * p.C + 32) * p.ldD, false);
            p.alloc("Bw", sizeof(cplx) * nb * p.C * p.ldS, false);
            p.alloc("Vw", sizeof(cplx) * nb * p.C * p.ldS, false);
            p.alloc("Zw", sizeof(cplx) * nb * p.C * p.ldS, false);
            p.alloc("tauw", sizeof(double) * nb * p.C);
            p.alloc("R2w", sizeof(cplx) * nb * p.C * p.C);
            p.alloc("Nw", sizeof(cplx) * nb * p.C * p.C);
            p.alloc("sv", sizeof(double) * (size_t)p.P * p.C);
            p.alloc("jsweeps", sizeof(int) * (size_t)p.P);
            if (d.kind == EMAGLS_KIND_EMAGLS && p.nOut > 32) {
                p.alloc("Ag", sizeof(double) * (size_t)p.nOut * p.nOut);
                p.alloc("Minv", sizeof(cplx) * (size_t)p.nOut * p.nOut);
            }
        } else {
        p.alloc("route", sizeof(int) * (size_t)p.P);
        p.alloc("Gy", esz(cb) * (size_t)p.S * p.S);                    // Gram matrix of conj(Y) (upper block triangle)
        p.sweep_persist = plan_persist_possible(p);
        p.synth_want = synth_enabled() && !cb && !p.custom_basis && !p.diffuse && d.kind != EMAGLS_KIND_EMA_SH &&
                       synth_sweep_supported((int)p.D, (int)d.nmics, p.simOrder + 1) && persist_sweep_supported((int)p.D, (int)d.nmics);
        plan_routes(p);
        plan_alloc_routes(p);
        p.alloc("sv", sizeof(double) * (size_t)p.P * p.C);
        p.alloc("jsweeps", sizeof(int) * (size_t)p.P);
        p.alloc("tauw", sizeof(double) * (size_t)p.P * p.C);
        p.alloc("R2w", sizeof(cplx) * (size_t)p.P * p.C * p.C);
        p.alloc("Nw", sizeof(cplx) * (size_t)p.P * p.C * p.C);
        p.alloc("Mw", sizeof(cplx) * ((size_t)p.P * p.C * p.C + 1024));
        p.alloc("cond_ok", sizeof(double) * (size_t)p.P);
        p.alloc("QT", esz(cb) * (size_t)(p.simOrder + 1) * p.C * p.ldD);
        if (getenv("EMAGLS_SWEEP_TIMING")) p.alloc("sweep_timing", sizeof(long long) * 16 * (size_t)p.P);
        }
    }
    if (d.kind == EMAGLS_KIND_FROM_ATF) {
        const int M = p.C;
        p.alloc("atf", sizeof(double) * (size_t)d.atf_taps * M * d.natf, false);
        p.alloc("atf_azi", sizeof(double) * d.natf, false);
        p.alloc("atf_zen", sizeof(double) * d.natf, false);
        p.alloc("cartB", sizeof(double) * 3 * std::max(d.natf, d.ndirs));
        p.alloc("match_idx", sizeof(int64_t) * p.Dm);
        p.alloc("match_dev", sizeof(double) * p.Dm);
        p.alloc("mean_dev", sizeof(double));
        p.alloc("colidx", sizeof(int64_t) * (size_t)p.Dm * M);
        p.ldD = round_up(p.Dm, 64);
        p.alloc("X", sizeof(cplx) * ((size_t)p.P * M + 32) * p.ldD);   // (+ 32 rows: the persistent sweep loads whole 32-row slabs)
        p.alloc("Z", sizeof(cplx) * ((size_t)p.P * M + 32) * p.ldD);
        // the per-bin M x M factors of the persistent sweep's form (Gram route, gramroute.hip): A_k = X_k X_k^H from the matched
        // ATF spectra themselves, M_k = V diag(s_reg / s) V^H
        p.alloc("Apk", sizeof(double) * (size_t)p.P * round_up(M * M, 64));
        p.alloc("Mw", sizeof(cplx) * ((size_t)p.P * M * M + 1024));
        p.alloc("cond_ok", sizeof(double) * (size_t)p.P);
        p.alloc("route", sizeof(int) * (size_t)p.P);
        p.gram_from = 1;   // every bin starts on the Gram route; a device-side conditioning flag moves the start up (plan_recover)
        p.alloc("Vws", sizeof(cplx) * (size_t)p.P * M * p.ldD);
        if (p.wide) p.alloc("Bw", sizeof(cplx) * (size_t)p.P * M * p.ldD);   // the matched ATF matrices again: the QR works in place
        p.alloc("sv", sizeof(double) * (size_t)p.P * M);
        p.alloc("jsweeps", sizeof(int) * (size_t)p.P);
        p.alloc("tauw", sizeof(double) * (size_t)p.P * M);
        p.alloc("R2w", sizeof(cplx) * (size_t)p.P * M * M);
        p.alloc("Nw", sizeof(cplx) * (size_t)p.P * M * M);
    }
    if (d.kind != EMAGLS_KIND_LS) {
        const int64_t Dh = (d.kind == EMAGLS_KIND_FROM_ATF) ? p.Dm : p.D;
        const int n_c = std::max(std::min(p.kcut0, p.P), 1);
        p.alloc("Hc", sizeof(cplx) * (size_t)2 * n_c * p.ldD);
        p.alloc("Habs", sizeof(double) * (size_t)2 * std::max(p.P - p.kcut0, 1) * p.ldD);
        p.alloc("W", sizeof(cplx) * (size_t)2 * p.P * p.C);
        if (magls_kind(d.kind) || d.kind == EMAGLS_KIND_FROM_ATF || p.wide) p.nWG = dense_sweep_nwg((int)Dh, p.C);
        p.nWG_dense = dense_sweep_nwg((int)Dh, p.C);
        // the persistent sweep keeps one workgroup per CU resident (142 KB of LDS each): it needs the shape to fit one XCD's
        // 32 CUs per design AND that many CUs on this device (a partitioned or CU-masked GPU takes the launch-per-bin form)
        p.sweep_persist = plan_persist_possible(p);
        if (magls_kind(d.kind) && p.sweep_persist) {
            p.alloc("Gm", sizeof(cplx) * ((size_t)p.C + 32) * p.ldD);      // Y_conj as complex [c][d] (+ 32 rows: whole-slab loads)
            p.alloc("Mw", sizeof(cplx) * ((size_t)p.P * p.C * p.C + 1024));
            p.alloc("cond_ok", sizeof(double) * (size_t)p.P);
        }
        p.alloc("ll", std::max(persist_sweep_ll_bytes((int)Dh, p.synth_want ? std::max(p.C, (int)d.nmics) : p.C),
                               p.synth_want ? reg_sweep_ll_bytes((int)Dh, (int)d.nmics) : (size_t)0));
        if (p.synth_want) p.alloc("sweep_args", sizeof(HalfSweepArgs));
        p.alloc("Wpart", sizeof(cplx) * (size_t)2 * std::max(p.nWG, p.nWG_dense) * 2 * p.C);
        p.out_rows = d.len;
    }
    p.out_cols = p.C;
    p.out_cplx = p.req_cplx && d.kind != EMAGLS_KIND_FROM_ATF;
    p.alloc("wL", (p.out_cplx ? sizeof(cplx) : sizeof(double)) * (size_t)p.out_rows * p.out_cols);
    p.alloc("wR", (p.out_cplx ? sizeof(cplx) : sizeof(double)) * (size_t)p.out_rows * p.out_cols);
    const auto t_setup2 = std::chrono::steady_clock::now();
    HIP_CHECK(hipStreamSynchronize(p.stream));
    if (trace_on()) {
        const auto t_setup3 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "emagls trace: plan setup: streams %.3f ms, %zu buffers in %zu slabs %.3f ms, final sync %.3f ms\n", ms(t_setup0, t_setup1), p.bufs.size(),
                p.slabs.size(), ms(t_setup1, t_setup2), ms(t_setup2, t_setup3));
        static std::atomic<int> shown{0};
        if (shown.fetch_add(1) % 64 == 0) {   // (every 64th plan: its largest buffers)
            std::vector<std::pair<size_t, std::string>> big;
            for (auto& kv : p.bufs) big.emplace_back(kv.second.bytes, kv.first);
            std::sort(big.rbegin(), big.rend());
            std::string line;
            for (size_t i = 0; i < big.size() && i < 12; ++i) line += " " + big[i].second + "=" + std::to_string(big[i].first >> 20);
            fprintf(stderr, "emagls trace: plan of %lld MB (sim order %d, S_h %d, hh_end %d); largest buffers (MB):%s\n", (long long)(p.total_bytes >> 20), p.simOrder, p.S_h, p.hh_end, line.c_str());
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pipelines
// ---------------------------------------------------------------------------------------------
// SH matrix on the HRIR grid, Cholesky-QR:  conj(Y) = Q R
void stage_hrir_basis(emagls_plan& p) {
    hipStream_t st = p.stream;
    const bool cb = p.cplx_basis;
    if (p.d.kind == EMAGLS_KIND_MAGLS_2D) {   // circular harmonics of the horizontal HRIR grid (getMagLsFilters2D.m:49)
        launch_ch_basis(p.d.order, (int)p.D, p.get<double>("hrir_azi"), cb, p.get("Ycm"), (int)p.ldD, st, !cb);
    } else if (!p.custom_basis) {
        launch_sh_coeff(p.simOrder, p.get<double>("sh_tab"), st);
        launch_sh_basis(p.simOrder, p.D, p.get<double>("hrir_azi"), p.get<double>("hrir_zen"), p.get<double>("sh_tab"), cb,
                        p.get("Ycm"), p.ldD, st);
    }
    launch_transpose_conj(p.get("Ycm"), p.D, p.S, p.ldD, p.get("Yc"), p.Dpad, p.ldS, cb, true, st);
    p.mark("sh_basis");
    launch_gram(p.get("Yc"), p.D, p.S, p.ldS, cb, p.get("Gp"), nullptr, p.get("R"), p.S, st);
    p.mark("gram_mfma");
    launch_cholesky(p.get("R"), p.S, cb, p.get<int>("flag"), st);
    p.mark("cholesky");
    if (p.wide) return;   // (no orthonormal factor: pinv(Y_conj) = Y conj((Y^T conj(Y))^-1), run_pinv_of_R)
    launch_qform(p.get("Yc"), p.get("R"), p.get("Rinv"), p.S, p.D, p.ldS, cb, p.get("Q"), st);
    p.mark("qform");
}

void stage_prologue(emagls_plan& p, int mode, const int64_t* didx, int64_t Dh) {
    hipStream_t st = p.stream;
    const emagls_design_desc& d = p.d;
    launch_twiddles(p.nfft, p.get("tw"), st);
    // group delay from the sum over ALL HRIR directions (lib/getEMagLsFilters.m:74-75)
    launch_hrir_grpdelay(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, d.ndirs, p.nfft, p.get("tw"),
                         p.get<double>("dirsum"), p.get<double>("grpd"), st);
    const int n_c = std::max(std::min(p.kcut0, p.P), 1);
    launch_hrir_fft(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, Dh, didx, p.nfft, p.get("tw"), p.get<double>("grpd"),
                    mode, std::min(p.kcut0, p.P), p.kcut0, p.get("Hc"), p.get<double>("Habs"), p.ldD, st);
    (void)n_c;
    if (p.diffuse && mode == 0)
        launch_hrir_fft(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, Dh, didx, p.nfft, p.get("tw"), p.get<double>("grpd"),
                        0, p.P, p.P, p.get("Hfull"), p.get<double>("Habs"), p.ldD, st);
    p.mark("hrir_prologue");
}

void record_sweep_event(emagls_plan& p, size_t i) {
    if (p.sweep_events.size() <= i) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        p.sweep_events.push_back(e);
    }
    HIP_CHECK(hipEventRecord(p.sweep_events[i], p.stream));
}

void run_pinv_of_R(emagls_plan& p) {
    // pinv(Y_conj) = conj(Q) Z_B, Z_B from the SVD of B = R with MATLAB's pinv tolerance
    hipStream_t st = p.stream;
    const bool cb = p.cplx_basis;
    if (p.wide) {   // 33..64 channels: the inverse of the Gram matrix, certified well conditioned on the device (wide.hip)
        launch_gram_inverse(p.get("R"), p.S, cb, p.get("Mg"), p.get<int>("flag"), st, p.has("Mgw") ? p.get("Mgw") : nullptr);
        launch_ypinv_gram(p.get("Ycm"), p.ldD, cb, p.get("Mg"), p.S, (int)p.D, p.get("Ypinv"), st);
        p.mark("pinv");
        return;
    }
    launch_widen(p.get("R"), p.S, cb, p.get("Rb"), p.ldS, p.C, p.S, /*transpose=*/true, /*upper_only=*/true, st);
    FactorArgs a{};
    a.S = p.S; a.C = p.C; a.ldS = p.ldS; a.kb0 = 0; a.P = 2;  // P=2: bin 0 is not a Nyquist bin
    a.Tn = nullptr; a.bn = nullptr; a.nOrders = 0;
    a.Xd = p.get<cplx>("Rb"); a.xd_stride = 0;
    a.reg_mode = 1; a.reg_c = 0.0; a.tol_dim = (double)std::max<int64_t>(p.D, p.C);
    a.Z = p.get<cplx>("Zb"); a.Vws = p.get<cplx>("Vws"); a.sv = p.get<double>("sv");
    a.Hq = nullptr; a.W = nullptr; a.ls_end = 0; a.sweeps_out = nullptr;
    a.tauw = p.get<double>("tauw"); a.R2w = p.get<cplx>("R2w"); a.Nw = p.get<cplx>("Nw");
    launch_factor(a, 1, true, st);
    launch_ypinv(p.get("Q"), p.ldS, cb, p.get("Zb"), p.ldS, (int)p.D, p.S, p.C, p.get("Ypinv"), p.ldD, st);
    p.mark("pinv");
}

void execute_ls(emagls_plan& p) {
    stage_hrir_basis(p);
    run_pinv_of_R(p);
    launch_ls_filters(p.get<double>("hL"), p.get<double>("hR"), p.d.nsamp, (int)p.D, p.get("Ypinv"), p.cplx_basis, p.ldD,
                      p.C, p.get("wL"), p.get("wR"), p.stream);
    p.mark("ls_filters");
}

// MagLS / MagLS-2D: everything before the sweep.  With the persistent sweep (the kernel of the array designs, one resident launch
// instead of one launch per bin) the operands are the same for every bin: G = Y_conj as complex [c][d] and
// M = (G^H G)^-1 = R^-1 R^-H from the Cholesky factor, since pinv(Y_conj) = conj(G) conj(M) for a full-rank basis.
void magls_pre_sweep(emagls_plan& p) {
    hipStream_t st = p.stream;
    const bool cb = p.cplx_basis;
    stage_hrir_basis(p);
    run_pinv_of_R(p);
    launch_conj_copy(p.get("Ycm"), p.get("Xc"), (int64_t)p.S * p.ldD, cb, st);  // Y_conj [c][d]
    stage_prologue(p, 0, nullptr, p.D);
    launch_ls_apply(p.get("Hc"), p.ldD, std::min(p.kcut0, p.P), p.get("Ypinv"), cb, p.ldD, (int)p.D, p.C, p.P, 0,
                    std::min(p.kcut0, p.P), p.get("W"), st);
    p.mark("ls_bins");
    if (p.sweep_persist) {
        launch_widen(p.get("Xc"), p.ldD, cb, p.get("Gm"), p.ldD, p.C, (int)p.D, false, false, st);
        launch_magls_m(p.get("R"), p.S, cb, p.P, p.get("Mw"), p.get<double>("cond_ok"), p.get<int>("flag"), st);
        p.mark("sweep_operands");
    }
}
void magls_post_sweep(emagls_plan& p) {
    hipStream_t st = p.stream;
    const bool cb = p.cplx_basis;
    if (p.diffuse)   // pwGrid is Y_conj for every bin
        launch_diffuse_constraint(p.get("W"), p.get("Xc"), cb, 0, 1, p.get("Hfull"), (int)p.D, p.C, p.ldD, p.P, st);
    // complex basis: getShFreqDomainConjugate (getMagLsFilters.m) / getChFreqDomainConjugate (getMagLsFilters2D.m:82-83)
    launch_filter_epilogue(p.get("W"), p.C, p.nfft, (int)p.d.len, p.get("tw"), p.get<double>("grpd"),
                           cb ? (p.d.kind == EMAGLS_KIND_MAGLS_2D ? 2 : 1) : 0, 0, 0,
                           p.out_cplx ? 1 : 0, p.get("wL"), p.get("wR"), st);
    p.mark("epilogue");
}
void emagls_run_sweep(emagls_plan& p);
void execute_magls(emagls_plan& p) {
    hipStream_t st = p.stream;
    const bool cb = p.cplx_basis;
    magls_pre_sweep(p);
    if (p.sweep_persist) {   // (eager / profiled executes; plan_execute captures the two halves around the sweep otherwise)
        emagls_run_sweep(p);
        magls_post_sweep(p);
        return;
    }
    DenseSweepArgs a{};
    a.D = (int)p.D; a.C = p.C; a.ldD = (int)p.ldD; a.P = p.P;
    a.X = p.get("Xc"); a.x_stride = 0; a.Zd = p.get("Ypinv"); a.z_stride = 0;
    a.Habs = p.get<double>("Habs"); a.ldH = p.ldD; a.kabs0 = p.kcut0;
    a.Wpart = p.get<cplx>("Wpart"); a.W = p.get<cplx>("W"); a.nWG = p.nWG; a.dpw = 0; a.kfirst = p.kcut0;
    p.sweep_launches = 0;
    for (int kb = p.kcut0; kb < p.P; ++kb) {
        if (p.prof_level >= 2) record_sweep_event(p, 2 * (size_t)p.sweep_launches);
        if (p.wide) launch_sweep_wide(a, kb, cb, st); else launch_sweep_dense(a, kb, cb, st);
        if (p.prof_level >= 2) record_sweep_event(p, 2 * (size_t)p.sweep_launches + 1);
        ++p.sweep_launches;
    }
    if (p.kcut0 < p.P) {
        if (p.wide) launch_sweep_wide_finalize(p.get("Wpart"), p.get("W"), p.nWG, p.C, p.P, p.P - 1, st);
        else launch_sweep_finalize(p.get("Wpart"), p.get("W"), p.nWG, p.C, p.P, p.P - 1, st);
    }
    p.mark("magls_sweep");
    magls_post_sweep(p);
}

// First bin of the Gram route: cond(B_k) is governed by the ratio of the lowest to the highest modal coefficient the C
// output channels can carry, |b_0 / b_n| ~ (2n+1)!! / (kr)^n with n = ceil(sqrt(C)) - 1; the route starts where that
// estimate falls below GRAM_COND_EST (the Jacobi kernel verifies cond < 10x that and asks for a re-run otherwise).  The
// route's error is eps cond^2 <= 2e-7 eps-relative at the verification limit 3e4, i.e. 2e-8 on M_k: two orders inside the
// 1e-6 parity tolerance.  With 3e3 the Householder route of the em32 design ends at 1 kHz (bin 21 of 513), where 16 orders are
// above the noise floor: 256 rows, the register tile with which its kernels fit next to a resident sweep workgroup.
constexpr double GRAM_COND_EST = 3.0e3;
double gram_cond_est() {
    if (const char* e = getenv("EMAGLS_GRAM_COND_EST")) return atof(e);   // (tests force the re-run path with a huge limit)
    return GRAM_COND_EST;
}
int emagls_gram_from(const emagls_plan& p) {
    if (!p.gram_route || p.d.mic_radius <= 0.0) return 0;
    if (const char* e = getenv("EMAGLS_GRAM_ROUTE")) if (e[0] == '0') return 0;
    int n = 0;
    if (p.d.kind == EMAGLS_KIND_EMA_CH) n = p.d.order;   // 2N+1 circular harmonics reach order N
    else while ((n + 1) * (n + 1) < p.C) ++n;
    if (n < 1) return 0;
    double dfact = 1.0;
    for (int i = 3; i <= 2 * n + 1; i += 2) dfact *= i;
    const double est_limit = gram_cond_est();
    const double kr_min = std::pow(dfact / est_limit, 1.0 / n);
    const double df = p.d.fs / p.nfft;
    const int kb = (int)std::ceil(kr_min * C_SOUND / (2.0 * kPi * p.d.mic_radius) / df);
    // (least-squares bins above the estimate take the route as well: W(k,:) = (H conj(G_k)) conj(M_k), gramroute.hip)
    const int from = std::max(kb, 1);
    return from < p.P ? from : 0;
}

// bins per Jacobi workgroup on the Gram route (warm start from the neighbouring bin: fewer rotations per bin, but the launch lasts as
// long as its longest run).  Round 5, config 3 through the job scheduler: a lone chunk of 20 designs on forked streams -- nothing else
// on the GPU, the launch on the critical path -- 2219 / 2231 sets/s with runs of 4, 2263 / 2301 with 2, 2277 / 2311 with single bins;
// chunks in flight next to each other (128 / 512 steps): 3474-3494 / 3537-3639 with 4, 3471-3497 / 3550-3597 with 2, 3375-3465 /
// 3544-3572 with 1.  So: single bins where the stages before the sweep are forked (latency mode), runs of 2 in lane groups.
int jacobi_run_length(const emagls_plan& p) {
    static const int forced = [] { const char* e = getenv("EMAGLS_JACOBI_RUN"); return e ? std::max(1, atoi(e)) : 0; }();
    if (forced) return forced;
    return (batch_ctx().n >= 2 && p.nstreams <= 1) ? 2 : 1;
}

// getEMagLsFiltersEMAinSH: the HRIR prologue, the array model, the per-direction rotations and G_k of every bin (kernels and derivation:
// emash.hip).  One stream.
void ema_sh_operands(emagls_plan& p) {
    const emagls_design_desc& d = p.d;
    const bool cb = p.cplx_basis;
    hipStream_t st = p.stream;
    const int M = (int)d.nmics, ldM = round_up(M, 64), N = d.order, nCh = 2 * N + 1, nOrd = p.simOrder + 1;
    const int ls_end = std::min(p.kcut0, p.P);
    const int npts = ema_sh_npts(p.C), ldP = round_up(npts, 64);
    const int64_t ldA = round_up((int64_t)(p.D + 1) * npts, 64);
    const int64_t g_stride = (int64_t)p.C * p.ldD;
    p.sync_used = 0;
    // ---- HRIR prologue
    launch_twiddles(p.nfft, p.get("tw"), st);
    launch_hrir_grpdelay(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, d.ndirs, p.nfft, p.get("tw"), p.get<double>("dirsum"),
                         p.get<double>("grpd"), st);
    launch_hrir_fft(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, p.D, nullptr, p.nfft, p.get("tw"), p.get<double>("grpd"), 0, ls_end,
                    p.kcut0, p.get("Hc"), p.get<double>("Habs"), p.ldD, st);
    if (p.diffuse)
        launch_hrir_fft(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, p.D, nullptr, p.nfft, p.get("tw"), p.get<double>("grpd"), 0, p.P,
                        p.P, p.get("Hfull"), p.get<double>("Habs"), p.ldD, st);
    p.mark("hrir_prologue");
    // ---- array model: E0 = J pinv(CH(micAzi)) Y_mic   (EMAinSH.m:66-82), b_n(kr)
    launch_sh_coeff(p.simOrder, p.get<double>("sh_tab"), st);
    launch_sh_basis(p.simOrder, M, p.get<double>("mic_azi"), p.get<double>("mic_zen"), p.get<double>("sh_tab"), cb, p.get("Ymic_cm"), M, st);
    launch_transpose_conj(p.get("Ymic_cm"), M, p.S, M, p.get("Ymic_rm"), M, p.ldS, cb, false, st);
    launch_ch_basis(N, M, p.get<double>("mic_azi"), cb, p.get("Ylo_c"), ldM, st);
    {
        FactorArgs a{};
        a.S = M; a.C = nCh; a.ldS = ldM; a.kb0 = 0; a.P = 2;
        a.Xd = p.get<cplx>("Ylo_c"); a.xd_stride = 0;
        a.reg_mode = 1; a.tol_dim = (double)std::max(M, nCh);
        a.Z = p.get<cplx>("Zlo"); a.Vws = p.get<cplx>("Vlo");
        a.tauw = p.get<double>("tau_lo"); a.R2w = p.get<cplx>("R2_lo"); a.Nw = p.get<cplx>("N_lo");
        launch_factor(a, 1, true, st);
    }
    launch_small_gemm(p.get("Zlo"), ldM, true, p.get("Ymic_rm"), p.ldS, cb, p.get("Ech"), p.ldS, cb, nCh, p.S, M, st);
    launch_sh_coeff(N, p.get<double>("sh_tab_lo"), st);
    launch_sh_basis(N, p.C, p.get<double>("nnm_azi"), p.get<double>("nnm_zen"), p.get<double>("sh_tab_lo"), cb, p.get("Ypts"), p.C, st);
    launch_ema_sh_e0(p.get("Ech"), (int)p.ldS, p.get("Ypts"), p.C, p.S, cb, p.get("E"), st);
    launch_modal_bn(p.simOrder, p.P, p.get<double>("kr"), 1.0, -1.0, p.get("bn"), nOrd, 1, st, p.get<int>("nvalid"));
    p.mark("array_model");
    // ---- per-direction SH rotations (EMAinSH.m:85-100)
    launch_rot_points(p.get<double>("hrir_azi"), p.get<double>("hrir_zen"), (int)p.D, npts, p.get<double>("rot_azi"), p.get<double>("rot_zen"), st);
    launch_sh_basis(N, (p.D + 1) * npts, p.get<double>("rot_azi"), p.get<double>("rot_zen"), p.get<double>("sh_tab_lo"), cb, p.get("Arot"), ldA, st);
    {
        const char* B = (const char*)p.get("Arot") + esz(cb) * (size_t)p.D * npts;   // the unrotated point set: columns D*npts..
        launch_widen(B, ldA, cb, p.get("Bc"), ldP, p.C, npts, false, false, st);
        if (p.C > 32) {
            // orders 5..7: pinv of the npts x C point matrix by wide_array.hip's QR + one-sided Jacobi (no clipping: the point set
            // resolves the order, nothing is dropped) -- Z comes out as pinv(B) [C][ldP] like the narrow factorisation's
            launch_wa_factor(p.get("Bc"), p.get("Vb"), npts, p.C, ldP, 1, 0.0, p.get<double>("tau_b"), p.get("R2_b"), p.get("N_b"), p.get<double>("sv"),
                             p.get<int>("jsweeps"), p.get("Zb"), st);
        } else {
        FactorArgs a{};
        a.S = npts; a.C = p.C; a.ldS = ldP; a.kb0 = 0; a.P = 2;
        a.Xd = p.get<cplx>("Bc"); a.xd_stride = 0;
        a.reg_mode = 1; a.tol_dim = (double)std::max(npts, p.C);
        a.Z = p.get<cplx>("Zb"); a.Vws = p.get<cplx>("Vb");
        a.tauw = p.get<double>("tau_b"); a.R2w = p.get<cplx>("R2_b"); a.Nw = p.get<cplx>("N_b");
        launch_factor(a, 1, true, st);
        }
    }
    launch_rot_from_points(p.get("Arot"), ldA, p.get("Zb"), ldP, p.C, npts, p.get<double>("hrir_zen"), (int)p.D, cb, p.get("Rot"), st);
    p.mark("sh_rotations");
    // ---- order terms of pwGrid.' on the horizontal projection of the grid, rotated per direction; G_k of every bin
    launch_sh_basis(p.simOrder, p.D, p.get<double>("hrir_azi"), p.get<double>("hrir_zen_eq"), p.get<double>("sh_tab"), cb, p.get("Ycm"), p.ldD, st);
    launch_transpose_conj(p.get("Ycm"), p.D, p.S, p.ldD, p.get("Yc"), p.Dpad, p.ldS, cb, true, st);
    launch_qt(p.get("Yc"), p.ldS, p.get("E"), p.ldS, (int)p.D, p.S, p.C, nOrd, cb, p.get("QT"), p.ldD, st);
    launch_qt_rotate(p.get("QT"), p.ldD, nOrd, p.C, N, (int)p.D, p.get("Rot"), cb, st);
    launch_dspace_g(p.get("QT"), p.ldD, cb, p.get("bn"), nOrd, (int)p.D, p.C, p.P, p.g0, p.get("G"), st, 0, -1);
    p.mark("order_terms+G");
}
// everything before the sweep, up to order 4 (32 channels: the tuned Gram-route kernels and the resident sweep)
void ema_sh_pre_sweep(emagls_plan& p) {
    ema_sh_operands(p);
    const bool cb = p.cplx_basis;
    hipStream_t st = p.stream;
    const int ls_end = std::min(p.kcut0, p.P);
    const int64_t g_stride = (int64_t)p.C * p.ldD;
    (void)cb;
    // ---- per-bin C x C matrices: Gram route for every bin
    const int ldK = round_up(p.C * p.C, 64), gf = 1, nb = p.P - 1;
    launch_gram_from_g(p.get("G"), g_stride, p.ldD, (int)p.D, p.C, gf, nb, p.g0, p.get<double>("Apk"), ldK, st);
    launch_gram_solve(p.get<double>("Apk"), ldK, p.C, gf, nb, SVD_REGUL_CONST, p.get("Mw"), p.get("R2w"), p.get<double>("sv"),
                      p.get<int>("route"), p.get<int>("jsweeps"), st);
    {
        FactorArgs fg{};
        fg.S = p.C; fg.C = p.C; fg.ldS = round_up(p.C, 64); fg.kb0 = gf; fg.P = p.P;
        fg.reg_mode = 0; fg.reg_c = SVD_REGUL_CONST;
        fg.sv = p.get<double>("sv"); fg.route = p.get<int>("route"); fg.status = p.get<int>("flag");
        fg.cond_limit = 10.0 * GRAM_COND_EST;
        fg.sweeps_out = p.get<int>("jsweeps");
        fg.tauw = p.get<double>("tauw"); fg.R2w = p.get<cplx>("R2w"); fg.Nw = p.get<cplx>("Nw"); fg.Mw = p.get<cplx>("Mw");
        fg.jrun = jacobi_run_length(p);
        launch_factor_jacobi_gram(fg, nb, st);
    }
    launch_cond_flags(p.get<double>("sv"), p.C, p.P, 1, p.get<double>("cond_ok"), st);
    p.mark("gram_route");
    // ---- least-squares bins
    if (ls_end > 1)
        launch_ls_gram(p.get("Hc"), p.ldD, ls_end, (const cplx*)p.get("G") - (int64_t)p.g0 * g_stride, g_stride, p.ldD, p.get("Mw"), (int)p.D, p.C,
                       p.P, 1, ls_end, p.get("W"), st);
    p.mark("ls_bins");
}

// The synthesising sweep needs the Gram-route bins only: M~_k of the swept bins, the start value W(k_cut-1,:) (a least-squares bin of the
// Gram route), |H|, the Chebyshev coefficients.  The Cholesky factor of the grid's SH Gram matrix and the whole orthonormal route of
// the ill-conditioned low bins (T_n, Householder QR, Jacobi, back-transform, their least-squares rows: bins 1 .. hh_end-1) feed the
// filters' rows, i.e. the epilogue AFTER the sweep -- 0.9 ms of the 3 ms a lone 20-design chunk spent before its sweep.  A batch
// therefore runs them next to the sweep on a stream of their own (batch_execute_lanes).  EMAGLS_DEFER_HH=0: everything before the sweep.
bool plan_defers_hh_route(const emagls_plan& p) {
    static const bool on = [] { const char* e = getenv("EMAGLS_DEFER_HH"); return !(e && e[0] == '0'); }();
    const int k0 = std::max(p.kcut0, 1);
    return on && p.synth && !p.diffuse && p.prof_level == 0 && p.d.kind != EMAGLS_KIND_EMA_SH && p.gram_from > 0 && p.hh_end > 1 && p.hh_end <= k0 - 1;
}
void emagls_pre_sweep(emagls_plan& p) {
    if (p.d.kind == EMAGLS_KIND_EMA_SH) { ema_sh_pre_sweep(p); return; }
    const emagls_design_desc& d = p.d;
    const bool cb = p.cplx_basis;
    const bool raw = d.kind == EMAGLS_KIND_EMAGLS2;
    const int M = (int)d.nmics;
    const int ldM = round_up(M, 64);
    // side streams shorten one design's critical path; with several designs in flight they only add queue
    // contention, so a plan can be restricted to its main stream (emagls_plan_set_streams / EMAGLS_STREAMS=1)
    if (p.nstreams >= 2) p.need_sides(p.nstreams);   // (no-op for a lane group: batch_lanes_part lends the batch's streams)
    hipStream_t s0 = p.stream, s1 = p.nstreams >= 2 ? p.side[0] : s0, s2 = p.nstreams >= 3 ? p.side[1] : s0;
    hipStream_t s3 = p.nstreams >= 4 ? p.side[2] : s0;   // the Gram route of the per-bin factors (needs Gy, E, b_n; not the Cholesky factor)
    const int nOrd = p.simOrder + 1;
    const int ls_end = std::min(p.kcut0, p.P);
    const int k0 = std::max(p.kcut0, 1);
    // routes (plan_routes): Householder bins [1, hh_end) on the orders 0..n_h, Gram-route bins [gf, P) on all orders
    const int gf = p.gram_from, hh_end = p.hh_end, Sh = p.S_h, ldSh = p.ldS_h, nOrdH = p.n_h + 1;
    const int ls_h = std::min(ls_end, hh_end);     // least-squares bins [1, ls_h) on the Householder route, [ls_h, ls_end) on the Gram route
    const int64_t g_stride = (int64_t)p.C * p.ldD;
    cplx* Gk = p.get<cplx>("G") - (int64_t)p.g0 * g_stride;   // indexed by kb
    const int phase = plan_defers_hh_route(p) ? p.pre_phase : 0;
    if (phase != 2) p.sync_used = 0;

    // The stages before the sweep as blocks.  Their data dependencies: array (mic SH matrix, E, b_n) <- nothing; prologue (HRIR
    // spectra) <- nothing; basis (Yc) <- nothing; gram (Gy, R) <- basis; chol <- gram; gterms (QT_n, G_k) <- basis, array;
    // rows (H conj(Q)) <- prologue, chol; gram_route (M_k of the Gram-route bins) <- gram, array; hh_route (QR + Jacobi of the
    // Householder-route bins) <- chol, array; flags <- gram_route, hh_route; back (Z_k, least-squares rows) <- flags, rows;
    // tail (least-squares rows of the Gram route, accurate Y_reg_inv) <- gterms, back.
    // order 0 issues them as three or four branches on the plan's streams (one design: shortest critical path).  Orders 1 and 2
    // are single-stream sequences for lane groups that run side by side (batch_execute_lanes): order 1 issues the kernels that
    // fill the chip first (HRIR transform, Gram matrix, G_k) and the latency-bound chains after them (Cholesky, per-bin
    // factors), order 2 the other way round -- two groups in the SAME order meet at the same kernels and add up their times,
    // two groups in complementary orders hide one's chains behind the other's bandwidth-bound kernels.
    const int order = (s1 == s0 && s2 == s0 && s3 == s0) ? p.stage_order : 0;
    hipEvent_t e_E = nullptr, e_Yc = nullptr, e_Gy = nullptr, e_R = nullptr;
    FactorArgs fa{};

    auto blk_array = [&] {
    // s1: array model  E = Y_mic (raw) or pinv(Y_mic(:,1:nOut)) Y_mic   (getSMAIRMatrix.m:101-102,119-121), b_n(kr)
    if (!p.custom_basis)
        launch_sh_basis(p.simOrder, M, p.get<double>("mic_azi"), p.get<double>("mic_zen"), p.get<double>("sh_tab"), cb,
                        p.get("Ymic_cm"), M, s1);
    launch_transpose_conj(p.get("Ymic_cm"), M, p.S, M, p.get("Ymic_rm"), M, p.ldS, cb, false, s1);
    if (raw) {
        launch_transpose_conj(p.get("Ymic_cm"), M, p.S, M, p.get("E"), M, p.ldS, cb, false, s1);  // E = Y_mic
    } else {
        if (d.kind == EMAGLS_KIND_EMA_CH)   // pinv(chFunction(order, micGridAziRad))  (getEMagLsFiltersEMAinCH.m:70)
            launch_ch_basis(d.order, M, p.get<double>("mic_azi"), cb, p.get("Ylo_c"), ldM, s1);
        else
            launch_widen(p.get("Ymic_cm"), M, cb, p.get("Ylo_c"), ldM, p.nOut, M, false, false, s1);
        FactorArgs a{};
        a.S = M; a.C = p.nOut; a.ldS = ldM; a.kb0 = 0; a.P = 2;
        a.Xd = p.get<cplx>("Ylo_c"); a.xd_stride = 0;
        a.reg_mode = 1; a.tol_dim = (double)std::max(M, p.nOut);
        a.Z = p.get<cplx>("Zlo"); a.Vws = p.get<cplx>("Vlo");
        a.tauw = p.get<double>("tau_lo"); a.R2w = p.get<cplx>("R2_lo"); a.Nw = p.get<cplx>("N_lo");
        launch_factor(a, 1, true, s1);
        launch_small_gemm(p.get("Zlo"), ldM, true, p.get("Ymic_rm"), p.ldS, cb, p.get("E"), p.ldS, cb, p.nOut, p.S, M, s1);
    }
    // bnAll = -sphModalCoeffs(simOrder, kr, 'rigid')   (getSMAIRMatrix.m:107)
    launch_modal_bn(p.simOrder, p.P, p.get<double>("kr"), 1.0, -1.0, p.get("bn"), nOrd, 1, s1, p.get<int>("nvalid"));
    if (p.synth)   // scaled modal terms of the Legendre series and pinv(Y_lo) as the real matrix Pm (identity: raw microphones)
        launch_synth_prepare(p.get("bn"), nOrd, p.P, p.get("bsc"), raw ? nullptr : p.get("Zlo"), ldM, p.C, M, p.get<int>("smap"), p.get<double>("Pm"), s1);
    e_E = p.next_sync_event();
    if (s1 != s0) HIP_CHECK(hipEventRecord(e_E, s1));
    };

    auto blk_prologue = [&] {
    // s2: HRIR prologue
        launch_twiddles(p.nfft, p.get("tw"), s2);
        launch_hrir_grpdelay(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, d.ndirs, p.nfft, p.get("tw"),
                             p.get<double>("dirsum"), p.get<double>("grpd"), s2);
        launch_hrir_fft(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, p.D, nullptr, p.nfft, p.get("tw"),
                        p.get<double>("grpd"), 0, ls_end, p.kcut0, p.get("Hc"), p.get<double>("Habs"), p.ldD, s2,
                        ls_end > 0 ? p.get<double>("HcT") : nullptr, round_up(4 * std::max(ls_end, 1), 64));
        if (p.diffuse)   // the target covariance needs the complex HRTFs of all bins (the sweep only keeps |H| above k_cut)
            launch_hrir_fft(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, p.D, nullptr, p.nfft, p.get("tw"),
                            p.get<double>("grpd"), 0, p.P, p.P, p.get("Hfull"), p.get<double>("Habs"), p.ldD, s2);
    };

    auto blk_basis = [&] {
    // s0: SH matrix of the HRIR grid
    if (!p.custom_basis)
        launch_sh_basis(p.simOrder, p.D, p.get<double>("hrir_azi"), p.get<double>("hrir_zen"), p.get<double>("sh_tab"), cb,
                        p.get("Ycm"), p.ldD, s0);
    launch_transpose_conj(p.get("Ycm"), p.D, p.S, p.ldD, p.get("Yc"), p.Dpad, p.ldS, cb, true, s0);
    p.mark("sh_basis");
    e_Yc = p.next_sync_event();
    if (s1 != s0) HIP_CHECK(hipEventRecord(e_Yc, s0));
    };
    auto blk_gram = [&] {
    // s0: Gram matrix Gy of conj(Y); its leading block (Householder-route orders) goes to R
    launch_gram(p.get("Yc"), p.D, p.S, p.ldS, cb, p.get("Gp"), p.get("Gy"), p.get("R"), Sh, s0);
    p.mark("gram_mfma");
    e_Gy = p.next_sync_event();
    if (s3 != s0) HIP_CHECK(hipEventRecord(e_Gy, s0));
    };
    auto blk_chol = [&] {
    launch_cholesky(p.get("R"), Sh, cb, p.get<int>("flag"), s0);
    p.mark("cholesky");
    e_R = p.next_sync_event();
    if (s2 != s0) HIP_CHECK(hipEventRecord(e_R, s0));
    };

    auto blk_gterms = [&] {
    // s1 (after the array model): order terms of pwGrid.' and G_k of every bin from g0 on -- needs only conj(Y) and E
    if (s1 != s0) HIP_CHECK(hipStreamWaitEvent(s1, e_Yc, 0));
    // (the synthesising sweep and its least-squares bins evaluate their operands themselves: neither order terms nor G_k)
    const int g_end = p.synth ? p.g0 : p.P;
    if (g_end > p.g0) {
    launch_qt(p.get("Yc"), p.ldS, p.get("E"), p.ldS, (int)p.D, p.S, p.C, nOrd, cb, p.get("QT"), p.ldD, s1);
    // (complex-arithmetic pipeline: G_k is still evaluated on the real order terms; circular-harmonic
    // channels would need their own channel transform and take the complex kernel)
    launch_dspace_g(p.get("QT"), p.ldD, cb, p.get("bn"), nOrd, (int)p.D, p.C, p.P, p.g0, p.get("G"), s1,
                    (cb && d.kind != EMAGLS_KIND_EMA_CH && !p.custom_basis) ? 1 : 0, raw ? -1 : (int)d.order, g_end);
    }
    };
    auto blk_rows = [&] {
    // s2 (after the prologue): the least-squares right-hand sides H conj(Q) of the Householder-route bins.  Q itself is never
    // formed: H conj(Q) is conj( conj(H conj(Yc)) R^-1 ), one D-long product and a row solve for the least-squares rows.
    if (s2 != s0) HIP_CHECK(hipStreamWaitEvent(s2, e_R, 0));
    if (hh_end > 1) {
        // (real basis: the rows are complex all the same, so R is widened to a complex copy for the row solves)
        if (!cb) launch_widen(p.get("R"), Sh, false, p.get("Rc"), Sh, Sh, Sh, false, /*upper_only=*/true, s2);
        launch_hy_conj_mfma(p.get<double>("HcT"), round_up(4 * std::max(ls_end, 1), 64), ls_end, p.get("Yc"), p.ldS, cb, (int)p.D, Sh,
                            p.get<double>("Hyp"), p.get("Hq"), ldSh, s2);
        // (also forms the inverses of R's diagonal blocks, which the ill-conditioned swept bins need: at least one row)
        launch_qform(p.get("Hq"), p.get(cb ? "R" : "Rc"), p.get(cb ? "Rinv" : "Rinvc"), Sh, 2 * (int64_t)std::max(ls_end, 1), ldSh, true, p.get("Hq"), s2);
    }
    };

    // s0: per-bin factors.  Gram route first (needs E, b_n, Gy): K matrices, one GEMM over the bins, direct inverses
    // (on a stream of its own with four streams: it does not need the Cholesky factor, the Householder route does)
    fa.S = Sh; fa.C = p.C; fa.ldS = ldSh; fa.kb0 = 1; fa.P = p.P;
    fa.Tn = p.get("Tn"); fa.bn = p.get<cplx>("bn"); fa.nOrders = nOrdH; fa.bn_stride = nOrd;
    fa.reg_mode = 0; fa.reg_c = SVD_REGUL_CONST;
    fa.Z = p.get<cplx>("Z");
    fa.Mw = p.get<cplx>("Mw");
    fa.Vws = p.get<cplx>("Vws"); fa.sv = p.get<double>("sv");
    fa.Hq = p.get<cplx>("Hq"); fa.ldHq = ldSh; fa.hq_estride = (int64_t)ls_end * ldSh; fa.ls_end = ls_h;
    fa.hq_conj = 1;
    fa.route = p.get<int>("route"); fa.status = p.get<int>("flag");
    fa.cond_limit = 10.0 * GRAM_COND_EST;   // (not the env override: the forced-estimate test must trip this check)
    fa.W = p.get<cplx>("W"); fa.sweeps_out = p.get<int>("jsweeps");
    fa.tauw = p.get<double>("tauw"); fa.R2w = p.get<cplx>("R2w"); fa.Nw = p.get<cplx>("Nw");
    // single-stream sequences, EMAGLS_JACOBI_PAIR=1: the two Jacobi steps (Gram-route bins, Householder-route bins) as ONE launch --
    // each lasts as long as its slowest bin (216 us) and on one stream they add up.  Off by default: with four batches in flight the
    // shorter chain changes nothing (three runs each, 20 / 128 steps: 1831-1904 / 2260-2394 merged, 1789-1952 / 2334-2498 not)
    const char* e_jp = getenv("EMAGLS_JACOBI_PAIR");
    const bool merge_jacobi = s3 == s0 && p.nb_gram > 0 && hh_end > 1 && e_jp && e_jp[0] == '1';
    FactorArgs fg_deferred{};
    auto blk_gram_route = [&] {
    if (s1 != s0) HIP_CHECK(hipStreamWaitEvent(s0, e_E, 0));
    if (s3 != s0) { HIP_CHECK(hipStreamWaitEvent(s3, e_Gy, 0)); HIP_CHECK(hipStreamWaitEvent(s3, e_E, 0)); }
    if (p.nb_gram > 0) {
        const int ldK = round_up(p.C * p.C, 64), ldCf = round_up(p.P, 64);
        launch_gram_kmat(p.get("Gy"), p.get("E"), p.S, p.ldS, p.C, nOrd, cb, p.get("Fg"), p.ldS, p.get<double>("Kmat"), ldK, s3);
        launch_gram_gemm(p.get("bn"), nOrd, p.P, gf, p.nb_gram, p.get<double>("Cf"), ldCf, p.get<double>("Kmat"), ldK, p.C,
                         p.get<double>("Apk"), ldK, s3);
        launch_gram_solve(p.get<double>("Apk"), ldK, p.C, gf, p.nb_gram, SVD_REGUL_CONST, p.get("Mw"), p.get("R2w"), p.get<double>("sv"),
                          p.get<int>("route"), p.get<int>("jsweeps"), s3);
        // bins in which the 1 % clipping is active (cond > 100) or the certificate failed: Jacobi SVD of the Gram matrix
        FactorArgs fg = fa;
        fg.kb0 = gf;
        const int64_t off = (int64_t)(gf - 1);
        fg.R2w = fa.R2w + off * p.C * p.C; fg.Mw = fa.Mw + off * p.C * p.C; fg.Nw = fa.Nw + off * p.C * p.C; fg.tauw = fa.tauw + off * p.C;
        // batches have workgroups to spare: a Jacobi workgroup walks a run of neighbouring bins, each warm-started from the
        // previous one (a third of the sweeps); a single design keeps one bin per workgroup (shortest critical path)
        fg.jrun = jacobi_run_length(p);
        if (merge_jacobi) { fg_deferred = fg; }   // (one launch with the Householder-route bins: blk_hh_route)
        else launch_factor_jacobi_gram(fg, p.nb_gram, s3);
        p.mark("gram_route");
    }
    };
    auto blk_hh_route = [&] {
    // Householder route: T_n of the orders 0..n_h, per-bin QR + Jacobi
    if (hh_end > 1) {
        launch_tn(p.get("R"), p.get("E"), Sh, p.C, p.ldS, nOrdH, cb, p.get("Tn"), ldSh, s0);
        p.mark("array_model+tn");
        launch_factor(fa, hh_end - 1, cb, s0, merge_jacobi ? (1 | 8) : 1);
        if (merge_jacobi) launch_factor_jacobi_pair(fg_deferred, p.nb_gram, fa, hh_end - 1, s0);
    }
    };
    auto blk_flags = [&] {
    // cond_ok[kb]: the cheap direction-space identity is accurate for this bin.  Only the other swept bins (and the
    // least-squares bins) need Z_k, i.e. the back-transform
    p.depend(s0, s3);   // (singular-value bounds of the Gram-route bins)
    launch_cond_flags(p.get<double>("sv"), p.C, p.P, hh_end, p.get<double>("cond_ok"), s0);
    fa.cond_ok = p.get<double>("cond_ok");
    p.mark("factor_qr_jacobi");
    };
    auto blk_back = [&] {
    // join s2 (Hq, spectra, group delays): back-transform + least-squares bins of the Householder route
    p.depend(s0, s2);
    if (hh_end > 1) launch_factor(fa, hh_end - 1, cb, s0, 2);
    p.mark("factor_back+ls_bins");
    };
    auto blk_tail = [&] {
    // join s1 (G)
    p.depend(s0, s1);
    // least-squares bins on the Gram route
    if (gf > 0 && gf < ls_end) {
        if (p.synth) {   // u(k) = H(k,:) conj(g_k) from the angles, then W(k,:) = (u Pm^T) conj(M_k) like the swept bins' rows
            launch_synth_ls(p.get("Hc"), p.ldD, ls_end, p.get("bsc"), synth_nord_pad(nOrd), p.get<double>("hrir_azi"), p.get<double>("hrir_zen"),
                            p.get<double>("mic_azi"), p.get<double>("mic_zen"), p.get<int>("smap"), (int)p.D, M, p.P, gf, ls_end, p.get("Usw"), s0);
            launch_synth_rows(p.get("Usw"), synth_ls_chunks((int)p.D), p.get("Pm"), p.get("Mw"), p.C, M, gf, ls_end, p.P, p.get("W"), s0);
        } else
        launch_ls_gram(p.get("Hc"), p.ldD, ls_end, Gk, g_stride, p.ldD, p.get("Mw"), (int)p.D, p.C, p.P, gf, ls_end, p.get("W"), s0);
    }
    // ill-conditioned swept bins (Householder route only): Y_reg_inv_k = conj(Q) Z_k = conj(Yc) (Z_k R^-H); the flagged bins'
    // Z rows are solved in place first
    if (hh_end > k0) {
        launch_zsolve_flagged(p.get("Z"), ldSh, p.get(cb ? "R" : "Rc"), p.get(cb ? "Rinv" : "Rinvc"), p.get<double>("cond_ok"), Sh, p.C,
                              hh_end, k0, s0);
        launch_yri_accurate(p.get("Yc"), p.ldS, cb, p.get("Z"), ldSh, p.get<double>("cond_ok"), (int)p.D, Sh, p.C, hh_end, k0,
                            p.get("Yri"), p.ldD, s0, p.custom_basis ? nullptr : p.get("Ycm"), p.ldD);
    }
    // synthesising sweep: the chain runs in the microphone domain on Mt_k = Pm^T M_k Pm, from the start value W(k0-1,:) Pm
    if (p.synth) launch_synth_mt(p.get("Mw"), p.get<double>("Pm"), p.C, M, k0, p.P, p.get("W"), p.get("Mt"), p.get("Winit"), s0);
    };

    if (phase == 2) {   // what the sweep did not need, on one stream: Cholesky factor, orthonormal route of the low bins, their rows
        blk_chol(); blk_rows(); blk_hh_route(); blk_flags(); blk_back();
        return;
    }
    launch_sh_coeff(p.simOrder, p.get<double>("sh_tab"), s0);
    if (phase == 1) {   // only what the sweep needs (forked like order 0 when the plan has side streams)
        if (s1 != s0) p.depend(s1, s0);
        if (s2 != s0) p.depend(s2, s0);
        // (tried: the least-squares bins' partial sums u(k) = H(k,:) conj(g_k) on the prologue's stream, off this path -- 2755-2813 against
        // 2802-2855 sets/s at 20 steps: the Gram route then queued behind the HRIR transform in one hardware queue)
        blk_array(); blk_prologue(); blk_basis(); blk_gram(); blk_gterms(); blk_gram_route();
        p.depend(s0, s2);   // (spectra of the least-squares bins, |H|)
        p.depend(s0, s3);   // (M_k of the Gram-route bins when that route has a stream of its own)
        blk_tail();
        p.mark("yri_operands");
        return;
    }
    if (order == 1) {          // bandwidth-bound kernels first
        blk_array(); blk_prologue(); blk_basis(); blk_gram(); blk_gterms();
        blk_chol(); blk_rows(); blk_gram_route(); blk_hh_route(); blk_flags(); blk_back(); blk_tail();
    } else if (order == 2) {   // latency-bound chains first
        blk_array(); blk_basis(); blk_gram(); blk_chol(); blk_gram_route(); blk_hh_route(); blk_flags();
        blk_prologue(); blk_rows(); blk_gterms(); blk_back(); blk_tail();
    } else {
        // ---- fork: three independent branches
        p.depend(s1, s0);
        p.depend(s2, s0);
        blk_array(); blk_prologue(); blk_basis(); blk_gram(); blk_chol(); blk_gterms(); blk_rows();
        blk_gram_route(); blk_hh_route(); blk_flags(); blk_back(); blk_tail();
    }
    p.mark("yri_operands");
}

HalfSweepArgs emagls_half_args(emagls_plan& p) {
    const int k0 = std::max(p.kcut0, 1);
    HalfSweepArgs a{};
    a.D = (int)p.D; a.C = p.C; a.ldD = (int)p.ldD; a.P = p.P;
    a.g_stride = (int64_t)p.C * p.ldD;
    if (p.d.kind == EMAGLS_KIND_FROM_ATF) {   // G_k = the matched ATF spectra of bin k, [kb][m][ldD]; bins below the Gram route: Y_reg_inv in Z
        a.D = (int)p.Dm;
        a.G = p.get<cplx>("X");
        a.Yri = p.get<cplx>("Z");
    } else if (magls_kind(p.d.kind)) {   // one operand for every bin (magls_pre_sweep)
        a.g_stride = 0;
        a.G = p.get<cplx>("Gm");
        a.Yri = a.G;             // (never read: every bin is well conditioned)
    } else {
    a.G = p.get<cplx>("G") - (int64_t)p.g0 * a.g_stride;    // indexed by kb (G starts at bin g0 <= k0)
    a.Yri = p.get<cplx>("Yri") - (int64_t)k0 * a.g_stride;
    }
    a.Mw = p.get<cplx>("Mw") - (int64_t)1 * p.C * p.C;      // factor stage stores bin kb at slot kb-1
    a.cond_ok = p.get<double>("cond_ok");
    a.Habs = p.get<double>("Habs"); a.ldH = p.ldD; a.kabs0 = p.kcut0;
    a.Wpart = p.get<cplx>("Wpart"); a.W = p.get<cplx>("W"); a.nWG = p.nWG_dense; a.kfirst = k0;
    a.ll = p.get<unsigned long long>("ll");
    a.abort_flag = p.get<int>("flag") + 1;
    a.skip_flag = magls_kind(p.d.kind) ? p.get<int>("flag") + 4 : nullptr;
    a.timing = p.has("sweep_timing") ? p.get<long long>("sweep_timing") : nullptr;
    const char* fg = getenv("EMAGLS_PERSIST_GLOBAL");
    a.force_global = (fg && fg[0] == '1') ? 1 : 0;
    static const int fetch_mode = [] { const char* e = getenv("EMAGLS_SWEEP_FETCH"); return e ? std::max(0, std::min(4, atoi(e))) : 0; }();
    a.fetch_mode = fetch_mode;
    static const long long wait_ticks = [] { const char* e = getenv("EMAGLS_SWEEP_WAIT_MS"); return (long long)(e ? std::max(1, atoi(e)) : 20) * 100000ll; }();
    a.wait_ticks = wait_ticks;
    if (p.synth) {   // the chain's channels are the microphones (sweep_synth.hip)
        const int M = (int)p.d.nmics;
        a.C = M;
        a.G = nullptr; a.Yri = nullptr;
        a.Mw = p.get<cplx>("Mt") - (int64_t)M * M;
        a.dir_azi = p.get<double>("hrir_azi"); a.dir_zen = p.get<double>("hrir_zen");
        a.mic_azi = p.get<double>("mic_azi"); a.mic_zen = p.get<double>("mic_zen");
        a.smap = p.get<int>("smap");
        a.bsc = p.get<cplx>("bsc"); a.nord_pad = synth_nord_pad(p.simOrder + 1);
        static const int split = [] { const char* e = getenv("EMAGLS_SYNTH_SPLIT"); return e ? atoi(e) : 67; }();
        a.synth_split = split;
        static const int prio = [] { const char* e = getenv("EMAGLS_SYNTH_PRIO"); return e ? std::max(0, std::min(5, atoi(e))) : 5; }();
        a.synth_prio = prio;
        a.Winit = p.get<cplx>("Winit"); a.U = p.get<cplx>("Usw");
    }
    return a;
}

// A persistent sweep needs all of its workgroups resident.  Two sweeps launched from different streams could each get a
// part of the CUs and wait for the rest forever (the kernels would give up after their time-out and report an error), so the
// sweeps of a device pass through one gate that counts workgroup slots per XCD: a sweep is launched behind as many of the
// earlier ones (oldest first, by their completion events) as it takes for everything that may still be running next to it to
// fit.  The register-resident form (sweep_reg.hip) takes ceil(n / 8) x nWG of the 96 slots of its kind an XCD has (3 workgroups
// per CU), so several of its launches run side by side; the slab forms (sweep_persist.hip, sweep_synth.hip) fill every CU's LDS
// and take the whole gate.  The sweep is therefore never part of a captured graph (plans and batches capture the stages before
// it).  The gate's state is per device and guarded by a mutex: plans of different host threads may sweep on the same GPU.
struct SweepGate {
    struct Entry { hipEvent_t ev; int slots; };
    struct State { std::deque<Entry> inflight; std::vector<hipEvent_t> pool; int capacity = 0; };
    static std::mutex& mutex() { static std::mutex m; return m; }
    static State& state() {   // (call with the mutex held)
        static std::map<int, State> per_device;
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        State& st = per_device[dev];
        if (st.capacity == 0) st.capacity = std::max(1, reg_sweep_slots_per_xcd());
        return st;
    }
    std::unique_lock<std::mutex> lock;
    hipStream_t st;
    int slots;
    // slots_per_xcd <= 0: the whole device
    SweepGate(hipStream_t s, int slots_per_xcd) : lock(mutex()), st(s), slots(0) {
        State& g = state();
        slots = slots_per_xcd <= 0 ? g.capacity : std::min(slots_per_xcd, g.capacity);
        static const bool serial = [] { const char* e = getenv("EMAGLS_SWEEP_SERIAL"); return e && e[0] == '1'; }();
        if (serial) slots = g.capacity;
        // finished launches no longer hold slots (anywhere in the queue: launches of different sizes finish out of order)
        for (auto it = g.inflight.begin(); it != g.inflight.end();) {
            if (hipEventQuery(it->ev) == hipSuccess) { g.pool.push_back(it->ev); it = g.inflight.erase(it); }
            else ++it;
        }
        (void)hipGetLastError();   // (hipErrorNotReady of the query is not an error)
        // Everything that has not FINISHED may still run next to this launch unless this launch waits for it -- also a launch that
        // an earlier one already waits for (it may not even have started: sweeps are enqueued behind the stages before them).  So
        // an entry stays in the queue, and counts for every later launch, until its event reports completion; this launch waits
        // for the oldest entries, as many as it takes for the rest to fit next to it.
        int held = 0;
        for (const Entry& e : g.inflight) held += e.slots;
        for (const Entry& e : g.inflight) {
            if (held + slots <= g.capacity) break;
            HIP_CHECK(hipStreamWaitEvent(st, e.ev, 0));
            held -= e.slots;
        }
    }
    ~SweepGate() {   // (the lock is held from the waits to the record: no other sweep can slip in between)
        try {
            State& g = state();
            hipEvent_t ev = nullptr;
            if (!g.pool.empty()) { ev = g.pool.back(); g.pool.pop_back(); }
            else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
            const hipError_t e = ev ? hipEventRecord(ev, st) : hipErrorUnknown;
            if (e == hipSuccess) g.inflight.push_back(Entry{ev, slots});
            else {   // (e.g. the stream belongs to another device than the calling thread's current one: the next sweep would not
                     // be ordered behind this one -- never silently)
                (void)hipGetLastError();
                fprintf(stderr, "emagls: the sweep gate's event could not be recorded (%s): persistent sweeps are no longer ordered\n",
                        hipGetErrorString(e));
            }
        } catch (...) {}
    }
};

// EMAGLS_SWEEP_REG=0: the synthesising sweep keeps its slab form (sweep_synth.hip) for every design; 2: the register-resident form for
// launches of any size (read at every launch)
static int reg_sweep_mode() { const char* e = getenv("EMAGLS_SWEEP_REG"); return e ? atoi(e) : 1; }
// does the register-resident form serve these `n` designs (all synthesising, of one shape) in one launch?  Up to 8 designs (one per
// XCD) the slab form is the faster one -- 5.4 us per bin on 29 CUs per design against 7.0 us on 22 -- and a launch of its own has the
// device to itself anyway; from 9 designs on the register-resident form wins (16 designs: 3.8 against 3.4 ms next to each other, but
// 32 designs in 5.2 ms and room for other kernels).
static bool reg_sweep_wanted(emagls_plan* const* plans, int n) {
    const int mode = reg_sweep_mode();
    static const int reg_min = [] { const char* e = getenv("EMAGLS_SWEEP_REG_MIN"); return e ? std::max(1, atoi(e)) : 9; }();   // (experiments)
    if (mode == 0 || n < 1 || (mode == 1 && n < reg_min)) return false;
    const emagls_plan& q = *plans[0];
    for (int j = 0; j < n; ++j) {
        const emagls_plan& p = *plans[j];
        if (!p.synth || p.synth_units < 1 || p.synth_units > reg_sweep_max_units() || p.D != q.D || !p.has("sweep_args")) return false;
    }
    return reg_sweep_fits((int)q.D, (int)q.d.nmics, q.synth_units, q.simOrder + 1, n);
}
// the argument blocks of a launch in device memory (stored again only when one of them changed)
static void reg_args_upload(const HalfSweepArgs* host, int n, void* dev, std::vector<char>& last, hipStream_t st) {
    const size_t bytes = sizeof(HalfSweepArgs) * (size_t)n;
    if (last.size() == bytes && memcmp(last.data(), host, bytes) == 0) return;
    store_sweep_args(host, n, static_cast<HalfSweepArgs*>(dev), st);
    last.assign(reinterpret_cast<const char*>(host), reinterpret_cast<const char*>(host) + bytes);
}

void emagls_run_sweep(emagls_plan& p) {
    hipStream_t s0 = p.stream;
    const int k0 = std::max(p.kcut0, 1);
    HalfSweepMulti m{};
    m.n = 1;
    m.a[0] = emagls_half_args(p);
    p.sweep_launches = 0;
    if (k0 < p.P && p.sweep_persist) {
        emagls_plan* self = &p;
        p.reg_sweep = reg_sweep_wanted(&self, 1);
        SweepGate gate(s0, p.reg_sweep ? reg_sweep_gate_cost((int)p.D, 1) : 0);
        launch_zero(p.get("ll"), p.bufs["ll"].bytes, s0);
        if (p.reg_sweep) reg_args_upload(&m.a[0], 1, p.get("sweep_args"), p.sweep_args_last, s0);
        if (p.prof_level >= 2) record_sweep_event(p, 0);
        if (p.reg_sweep) launch_sweep_reg(p.get<HalfSweepArgs>("sweep_args"), m.a[0], 1, s0);
        else if (p.synth) launch_sweep_synth(m, s0); else launch_sweep_persist(m, s0);
        if (p.prof_level >= 2) record_sweep_event(p, 1);
        p.sweep_launches = 1;
    } else if (k0 < p.P) {
        for (int kb = k0; kb < p.P; ++kb) {
            if (p.prof_level >= 2) record_sweep_event(p, 2 * (size_t)p.sweep_launches);
            launch_sweep_half(m, kb, s0);
            if (p.prof_level >= 2) record_sweep_event(p, 2 * (size_t)p.sweep_launches + 1);
            ++p.sweep_launches;
        }
        launch_sweep_half_finalize(m, p.P - 1, s0);
    }
    p.mark("magls_sweep");
}

void emagls_post_sweep(emagls_plan& p) {
    const bool cb = p.cplx_basis;
    const bool raw = p.d.kind == EMAGLS_KIND_EMAGLS2;
    const int conj_mode = !p.req_cplx || raw ? 0 : (p.d.kind == EMAGLS_KIND_EMA_CH ? 2 : 1);   // Hermitian mirror / SH rule / CH rule
    if (p.synth) {   // the chain stored the microphone-domain totals u(k): W(k,:) = (u(k) Pm^T) conj(M_k) for the swept bins
        const emagls_plan& g = p.geo_from ? *p.geo_from : p;   // (geometry-sharing batches: plan 0's Pm and M_k)
        launch_synth_rows(p.get("Usw"), 1, g.bufs.at("Pm").p, g.bufs.at("Mw").p, p.C, (int)p.d.nmics, std::max(p.kcut0, 1), p.P, p.P, p.get("W"), p.stream,
                          p.geo_from != nullptr);
    }
    if (p.diffuse)   // (in the real-arithmetic pipeline W is still W_r here: the rendered HRTFs W G are the same in either basis)
        launch_diffuse_constraint(p.get("W"), p.get("G"), true, (int64_t)p.C * p.ldD, p.g0, p.get("Hfull"), (int)p.D, p.C, p.ldD, p.P,
                                  p.stream);
    if (p.real_internal && !raw) launch_sh_rows_to_complex(p.get("W"), p.C, 2 * p.P, (int)p.d.order, p.stream);   // W_c = W_r T_N
    (void)cb;
    launch_filter_epilogue(p.get("W"), p.C, p.nfft, (int)p.d.len, p.get("tw"), p.get<double>("grpd"), conj_mode, 1, 0,
                           p.out_cplx ? 1 : 0, p.get("wL"), p.get("wR"), p.stream);
    p.mark("epilogue");
}

// getEMagLsFiltersEMAinSH at orders 5..7 (36 / 49 / 64 channels; lib/getEMagLsFiltersEMAinSH.m:66-143): the per-direction rotations leave
// no common S-space, so pwGrid_k.' = G_k (D x C) is factored itself -- Householder QR, one-sided Jacobi on the triangular factor, 1 %
// clipping, back-transform: Y_reg_inv_k directly (wide_array.hip with Q = I, the form FromAtf takes above 32 microphones) --, then
// the least-squares bins and one sweep launch per bin.
void execute_ema_sh_wide(emagls_plan& p) {
    hipStream_t st = p.stream;
    const int nb = p.P - 1, ls_end = std::min(p.kcut0, p.P), k0 = std::max(p.kcut0, 1);
    const int64_t g_stride = (int64_t)p.C * p.ldD;
    ema_sh_operands(p);                                   // G_k of the bins 1 .. P-1 (g0 = 1)
    cplx* G = p.get<cplx>("G");
    cplx* Yri = p.get<cplx>("Yri");
    HIP_CHECK(hipMemcpyAsync(p.get("Bw"), G, sizeof(cplx) * (size_t)nb * g_stride, hipMemcpyDeviceToDevice, st));   // (the QR works in place)
    launch_wa_factor(p.get("Bw"), p.get("Vw"), (int)p.D, p.C, (int)p.ldD, nb, SVD_REGUL_CONST, p.get<double>("tauw"), p.get("R2w"), p.get("Nw"),
                     p.get<double>("sv") + p.C, p.get<int>("jsweeps") + 1, Yri, st);
    p.mark("factor_bins");
    launch_wa_ls(p.get("Hc"), p.ldD, ls_end, Yri, p.ldD, (int)p.D, p.C, p.P, 1, ls_end, p.get("W"), st);
    p.mark("ls_bins");
    DenseSweepArgs a{};
    a.D = (int)p.D; a.C = p.C; a.ldD = (int)p.ldD; a.P = p.P;
    a.X = G - g_stride; a.x_stride = g_stride;            // (indexed by kb: bin 1 at the buffer's start)
    a.Zd = Yri - g_stride; a.z_stride = g_stride;
    a.Habs = p.get<double>("Habs"); a.ldH = p.ldD; a.kabs0 = p.kcut0;
    a.Wpart = p.get<cplx>("Wpart"); a.W = p.get<cplx>("W"); a.nWG = p.nWG; a.dpw = 0; a.kfirst = k0;
    p.sweep_launches = 0;
    for (int kb = k0; kb < p.P; ++kb) { launch_sweep_wide(a, kb, true, st); ++p.sweep_launches; }
    if (k0 < p.P) launch_sweep_wide_finalize(p.get("Wpart"), p.get("W"), p.nWG, p.C, p.P, p.P - 1, st);
    p.mark("magls_sweep");
    launch_filter_epilogue(p.get("W"), p.C, p.nfft, (int)p.d.len, p.get("tw"), p.get<double>("grpd"), p.req_cplx ? 1 : 0, 1, 0,
                           p.out_cplx ? 1 : 0, p.get("wL"), p.get("wR"), st);
    p.mark("epilogue");
}

// eMagLS / eMagLS2 with 33..64 channels: wide_array.hip.  One stream, every bin on the S-space route.
void execute_emagls_wide(emagls_plan& p) {
    if (p.d.kind == EMAGLS_KIND_EMA_SH) { execute_ema_sh_wide(p); return; }
    const emagls_design_desc& d = p.d;
    hipStream_t st = p.stream;
    const bool raw = d.kind == EMAGLS_KIND_EMAGLS2;
    const int M = (int)d.nmics, nOrd = p.simOrder + 1, nb = p.P - 1;
    const int ls_end = std::min(p.kcut0, p.P), k0 = std::max(p.kcut0, 1);
    const int64_t g_stride = (int64_t)p.C * p.ldD;
    // ---- SH matrices, array model, modal terms  (geo: skipped when the plan keeps them from its last run on these grids)
    const bool geo = !p.geo_skip;
    if (geo) {
    launch_sh_coeff(p.simOrder, p.get<double>("sh_tab"), st);
    launch_sh_basis(p.simOrder, p.D, p.get<double>("hrir_azi"), p.get<double>("hrir_zen"), p.get<double>("sh_tab"), false, p.get("Ycm"), p.ldD, st);
    launch_transpose_conj(p.get("Ycm"), p.D, p.S, p.ldD, p.get("Yc"), p.Dpad, p.ldS, false, true, st);
    launch_sh_basis(p.simOrder, M, p.get<double>("mic_azi"), p.get<double>("mic_zen"), p.get<double>("sh_tab"), false, p.get("Ymic_cm"), M, st);
    if (raw) {
        launch_transpose_conj(p.get("Ymic_cm"), M, p.S, M, p.get("E"), M, p.ldS, false, false, st);   // E = Y_mic
    } else if (p.nOut <= 32) {
        const int ldM = round_up(M, 64);
        launch_transpose_conj(p.get("Ymic_cm"), M, p.S, M, p.get("Ymic_rm"), M, p.ldS, false, false, st);
        launch_widen(p.get("Ymic_cm"), M, false, p.get("Ylo_c"), ldM, p.nOut, M, false, false, st);
        FactorArgs a{};
        a.S = M; a.C = p.nOut; a.ldS = ldM; a.kb0 = 0; a.P = 2;
        a.Xd = p.get<cplx>("Ylo_c"); a.xd_stride = 0;
        a.reg_mode = 1; a.tol_dim = (double)std::max(M, p.nOut);
        a.Z = p.get<cplx>("Zlo"); a.Vws = p.get<cplx>("Vlo");
        a.tauw = p.get<double>("tau_lo"); a.R2w = p.get<cplx>("R2_lo"); a.Nw = p.get<cplx>("N_lo");
        launch_factor(a, 1, true, st);
        launch_small_gemm(p.get("Zlo"), ldM, true, p.get("Ymic_rm"), p.ldS, false, p.get("E"), p.ldS, false, p.nOut, p.S, M, st);
    } else {
        // pinv(Y_lo) = (Y_lo^T Y_lo)^-1 Y_lo^T: the M x nOut SH matrix of the microphone grid has full column rank and is well
        // conditioned for any array that resolves the order (certified on the device like the SH Gram matrix of wide.hip)
        launch_wa_lo_gram(p.get("Ymic_cm"), M, p.nOut, p.get<double>("Ag"), st);
        launch_cholesky(p.get("Ag"), p.nOut, false, p.get<int>("flag"), st);
        launch_gram_inverse(p.get("Ag"), p.nOut, false, p.get("Minv"), p.get<int>("flag"), st);
        launch_wa_e(p.get("Ymic_cm"), M, p.nOut, p.S, p.get("Minv"), p.get("E"), (int)p.ldS, st);
    }
    launch_modal_bn(p.simOrder, p.P, p.get<double>("kr"), 1.0, -1.0, p.get("bn"), nOrd, 1, st, p.get<int>("nvalid"));
    }
    p.mark("array_model");
    // ---- HRIR prologue
    launch_twiddles(p.nfft, p.get("tw"), st);
    launch_hrir_grpdelay(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, d.ndirs, p.nfft, p.get("tw"), p.get<double>("dirsum"), p.get<double>("grpd"), st);
    launch_hrir_fft(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, p.D, nullptr, p.nfft, p.get("tw"), p.get<double>("grpd"), 0, ls_end, p.kcut0,
                    p.get("Hc"), p.get<double>("Habs"), p.ldD, st);
    if (p.diffuse) {   // the covariance constraint's target: the time-aligned complex HRTFs of every bin; G starts at bin 1 here
        launch_hrir_fft(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, p.D, nullptr, p.nfft, p.get("tw"), p.get<double>("grpd"), 0, p.P, p.P,
                        p.get("Hfull"), p.get<double>("Habs"), p.ldD, st);
        p.g0 = 1;
    }
    p.mark("hrir_prologue");
    // ---- conj(Y) = Q R, order terms T_n = R(:,blk_n) E(:,blk_n)^T and QT_n, G_k of every solved bin
    if (geo) {
    launch_gram(p.get("Yc"), p.D, p.S, p.ldS, false, p.get("Gp"), nullptr, p.get("R"), p.S, st);
    launch_cholesky(p.get("R"), p.S, false, p.get<int>("flag"), st);
    launch_qform(p.get("Yc"), p.get("R"), p.get("Rinv"), p.S, p.D, p.ldS, false, p.get("Q"), st);
    launch_tn(p.get("R"), p.get("E"), p.S, p.C, (int)p.ldS, nOrd, false, p.get("Tn"), p.ldS, st);
    launch_qt(p.get("Yc"), p.ldS, p.get("E"), p.ldS, (int)p.D, p.S, p.C, nOrd, false, p.get("QT"), p.ldD, st);
    launch_dspace_g(p.get("QT"), p.ldD, false, p.get("bn"), nOrd, (int)p.D, p.C, p.P, 1, p.get("G"), st, 0, raw ? -1 : (int)d.order);
    p.mark("order_terms+G");
    // ---- per-bin factors (bins 1 .. P-1) and Y_reg_inv
    launch_wa_assemble(p.get("Tn"), p.get("bn"), nOrd, p.S, p.C, (int)p.ldS, p.P, 1, nb, p.get("Bw"), st);
    launch_wa_factor(p.get("Bw"), p.get("Vw"), p.S, p.C, (int)p.ldS, nb, SVD_REGUL_CONST, p.get<double>("tauw"), p.get("R2w"), p.get("Nw"),
                     p.get<double>("sv") + p.C, p.get<int>("jsweeps") + 1, p.get("Zw"), st);
    launch_wa_yri(p.get("Q"), p.ldS, p.get("Zw"), p.S, p.C, (int)p.ldS, (int)p.D, p.ldD, nb, p.get("Yri"), st);
    }
    p.mark("factor_bins");
    // ---- least-squares bins, sweep (G and Yri start at bin 1)
    launch_wa_ls(p.get("Hc"), p.ldD, ls_end, p.get("Yri"), p.ldD, (int)p.D, p.C, p.P, 1, ls_end, p.get("W"), st);
    p.mark("ls_bins");
    DenseSweepArgs a{};
    a.D = (int)p.D; a.C = p.C; a.ldD = (int)p.ldD; a.P = p.P;
    a.X = p.get<cplx>("G") - g_stride; a.x_stride = g_stride;
    a.Zd = p.get<cplx>("Yri") - g_stride; a.z_stride = g_stride;
    a.Habs = p.get<double>("Habs"); a.ldH = p.ldD; a.kabs0 = p.kcut0;
    a.Wpart = p.get<cplx>("Wpart"); a.W = p.get<cplx>("W"); a.nWG = p.nWG; a.dpw = 0; a.kfirst = k0;
    p.sweep_launches = 0;
    for (int kb = k0; kb < p.P; ++kb) { launch_sweep_wide(a, kb, true, st); ++p.sweep_launches; }
    if (k0 < p.P) launch_sweep_wide_finalize(p.get("Wpart"), p.get("W"), p.nWG, p.C, p.P, p.P - 1, st);
    p.mark("magls_sweep");
    emagls_post_sweep(p);
}

void execute_emagls(emagls_plan& p) {
    if (p.wide) { execute_emagls_wide(p); return; }
    emagls_pre_sweep(p);
    emagls_run_sweep(p);
    emagls_post_sweep(p);
}

// ---- getEMagLsFiltersFromAtf on the persistent sweep --------------------------------------------------------------------------
// pwGrid_k = atfsMatched(k,:,:) (M x Dm) is given, not modelled (FromAtf.m:100-104): G_k = X_k, its M x M Gram matrix from G_k
// itself (the EMAinSH route), M_k by the direct inverse / the Jacobi SVD of the Gram matrix.  Measured ATFs can be arbitrarily
// ill-conditioned at low frequencies: the device check of the Gram route (cond < 3e4) raises the status flag with the highest
// offending bin, the host moves the route's start behind it (plan_recover) and the bins below take the dense route
// (Householder QR + Jacobi SVD of X_k itself), whose Y_reg_inv the sweep reads directly (cond_ok = 0).
// What depends on the HRIR set of the subject, and what only on the grids and the ATFs (shared by a batch of subjects):
void from_atf_subject_stage(emagls_plan& p) {   // grid matching (cheap; the prologue needs the match) + HRIR prologue
    hipStream_t st = p.stream;
    const emagls_design_desc& d = p.d;
    if (p.hrir_smaller)
        launch_grid_match(p.get<double>("hrir_azi"), p.get<double>("hrir_zen"), d.ndirs, p.get<double>("atf_azi"),
                          p.get<double>("atf_zen"), d.natf, p.get<double>("cartB"), p.get<int64_t>("match_idx"),
                          p.get<double>("match_dev"), p.get<double>("mean_dev"), st);
    else
        launch_grid_match(p.get<double>("atf_azi"), p.get<double>("atf_zen"), d.natf, p.get<double>("hrir_azi"),
                          p.get<double>("hrir_zen"), d.ndirs, p.get<double>("cartB"), p.get<int64_t>("match_idx"),
                          p.get<double>("match_dev"), p.get<double>("mean_dev"), st);
    launch_atf_colidx(p.hrir_smaller ? p.get<int64_t>("match_idx") : nullptr, p.Dm, p.C, p.get<int64_t>("colidx"), st);
    p.mark("grid_match");
    stage_prologue(p, 1, p.hrir_smaller ? nullptr : p.get<int64_t>("match_idx"), p.Dm);
}
void from_atf_shared_stage(emagls_plan& p) {    // ATF spectra on the matched directions and the per-bin factors
    hipStream_t st = p.stream;
    const emagls_design_desc& d = p.d;
    const int M = p.C, gf = p.gram_from, nb = gf > 0 ? p.P - gf : 0;
    const int64_t g_stride = (int64_t)M * p.ldD;
    const int ls_end = std::min(p.kcut0, p.P);
    launch_real_fft_gather(p.get<double>("atf"), d.atf_taps, (int64_t)M * p.Dm, p.get<int64_t>("colidx"), p.nfft, p.get("tw"),
                           p.get("X"), g_stride, p.Dm, p.ldD, st);
    p.mark("atf_fft");
    if (nb > 0) {
        const int ldK = round_up(M * M, 64);
        launch_gram_from_g(p.get("X"), g_stride, p.ldD, (int)p.Dm, M, gf, nb, 0, p.get<double>("Apk"), ldK, st);
        launch_gram_solve(p.get<double>("Apk"), ldK, M, gf, nb, SVD_REGUL_CONST, p.get("Mw"), p.get("R2w"), p.get<double>("sv"),
                          p.get<int>("route"), p.get<int>("jsweeps"), st);
        FactorArgs fg{};
        fg.S = M; fg.C = M; fg.ldS = round_up(M, 64); fg.kb0 = gf; fg.P = p.P;
        fg.reg_mode = 0; fg.reg_c = SVD_REGUL_CONST;
        fg.sv = p.get<double>("sv"); fg.route = p.get<int>("route"); fg.status = p.get<int>("flag");
        fg.cond_limit = 10.0 * GRAM_COND_EST;
        fg.sweeps_out = p.get<int>("jsweeps");
        const int64_t off = (int64_t)(gf - 1);   // (gram_solve stores bin kb at slot kb - 1; the Jacobi kernel indexes from its first bin)
        fg.tauw = p.get<double>("tauw") + off * M; fg.R2w = p.get<cplx>("R2w") + off * M * M; fg.Nw = p.get<cplx>("Nw") + off * M * M;
        fg.Mw = p.get<cplx>("Mw") + off * M * M;
        fg.jrun = 1;
        launch_factor_jacobi_gram(fg, nb, st);
    }
    launch_cond_flags(p.get<double>("sv"), M, p.P, 1, p.get<double>("cond_ok"), st);   // 1 for every bin ...
    const int dense_end = gf > 0 ? gf : p.P;   // bins [1, dense_end) on the dense route
    if (dense_end > 1) {
        FactorArgs a{};
        a.S = (int)p.Dm; a.C = M; a.ldS = (int)p.ldD; a.kb0 = 1; a.P = p.P;
        a.Xd = p.get<cplx>("X"); a.xd_stride = g_stride;
        a.reg_mode = 0; a.reg_c = SVD_REGUL_CONST;
        a.Z = p.get<cplx>("Z"); a.Vws = p.get<cplx>("Vws"); a.sv = p.get<double>("sv");
        a.Hq = p.get<cplx>("Hc"); a.ldHq = p.ldD; a.hq_estride = (int64_t)ls_end * p.ldD; a.ls_end = std::min(ls_end, dense_end);
        a.W = p.get<cplx>("W"); a.sweeps_out = p.get<int>("jsweeps");
        a.tauw = p.get<double>("tauw"); a.R2w = p.get<cplx>("R2w"); a.Nw = p.get<cplx>("Nw");
        launch_factor(a, dense_end - 1, true, st);
        launch_zero(p.get<double>("cond_ok"), sizeof(double) * (size_t)dense_end, st);   // ... but the dense-route ones: the sweep reads their Y_reg_inv
    }
    p.mark("factor_bins");
}
// least-squares bins on the Gram route with the operands of `sh` (the plan itself, or the plan whose ATF side a batch shares)
void from_atf_ls_rows(emagls_plan& p, emagls_plan& sh, hipStream_t st) {
    const int ls_end = std::min(p.kcut0, p.P), gf = sh.gram_from;
    if (gf > 0 && gf < ls_end)
        launch_ls_gram(p.get("Hc"), p.ldD, ls_end, sh.get("X"), (int64_t)p.C * p.ldD, p.ldD, sh.get("Mw"), (int)p.Dm, p.C, p.P, gf, ls_end,
                       p.get("W"), st);
}
void from_atf_pre_sweep(emagls_plan& p) {
    from_atf_subject_stage(p);
    from_atf_shared_stage(p);
    from_atf_ls_rows(p, p, p.stream);
    p.mark("ls_bins");
}
void from_atf_post_sweep(emagls_plan& p) {
    launch_filter_epilogue(p.get("W"), p.C, p.nfft, (int)p.d.len, p.get("tw"), p.get<double>("grpd"), 0, 1, 1, 0, p.get("wL"),
                           p.get("wR"), p.stream);
    p.mark("epilogue");
}

// FromAtf with 33..64 microphones: pwGrid_k.' = X_k.' (Dm x M) is its own "S-space" (Q = I), so wide_array.hip's per-bin kernels --
// Householder QR, one-sided Jacobi, back-transform -- give Y_reg_inv_k directly; one sweep launch per bin (lib/getEMagLsFiltersFromAtf.m:97-120).
void execute_from_atf_wide(emagls_plan& p) {
    hipStream_t st = p.stream;
    const emagls_design_desc& d = p.d;
    const int M = p.C, nb = p.P - 1;
    const int64_t g_stride = (int64_t)M * p.ldD;
    if (p.hrir_smaller)
        launch_grid_match(p.get<double>("hrir_azi"), p.get<double>("hrir_zen"), d.ndirs, p.get<double>("atf_azi"),
                          p.get<double>("atf_zen"), d.natf, p.get<double>("cartB"), p.get<int64_t>("match_idx"),
                          p.get<double>("match_dev"), p.get<double>("mean_dev"), st);
    else
        launch_grid_match(p.get<double>("atf_azi"), p.get<double>("atf_zen"), d.natf, p.get<double>("hrir_azi"),
                          p.get<double>("hrir_zen"), d.ndirs, p.get<double>("cartB"), p.get<int64_t>("match_idx"),
                          p.get<double>("match_dev"), p.get<double>("mean_dev"), st);
    launch_atf_colidx(p.hrir_smaller ? p.get<int64_t>("match_idx") : nullptr, p.Dm, M, p.get<int64_t>("colidx"), st);
    p.mark("grid_match");
    stage_prologue(p, 1, p.hrir_smaller ? nullptr : p.get<int64_t>("match_idx"), p.Dm);
    launch_real_fft_gather(p.get<double>("atf"), d.atf_taps, (int64_t)M * p.Dm, p.get<int64_t>("colidx"), p.nfft, p.get("tw"),
                           p.get("X"), g_stride, p.Dm, p.ldD, st);
    p.mark("atf_fft");
    cplx* X = p.get<cplx>("X");
    cplx* Z = p.get<cplx>("Z");
    HIP_CHECK(hipMemcpyAsync(p.get("Bw"), X + g_stride, sizeof(cplx) * (size_t)nb * g_stride, hipMemcpyDeviceToDevice, st));
    launch_wa_factor(p.get("Bw"), p.get("Vws"), (int)p.Dm, M, (int)p.ldD, nb, SVD_REGUL_CONST, p.get<double>("tauw"), p.get("R2w"), p.get("Nw"),
                     p.get<double>("sv") + M, p.get<int>("jsweeps") + 1, Z + g_stride, st);
    p.mark("factor_bins");
    const int ls_end = std::min(p.kcut0, p.P);
    launch_wa_ls(p.get("Hc"), p.ldD, ls_end, Z + g_stride, p.ldD, (int)p.Dm, M, p.P, 1, ls_end, p.get("W"), st);
    p.mark("ls_bins");
    DenseSweepArgs a{};
    a.D = (int)p.Dm; a.C = M; a.ldD = (int)p.ldD; a.P = p.P;
    a.X = X; a.x_stride = g_stride; a.Zd = Z; a.z_stride = g_stride;
    a.Habs = p.get<double>("Habs"); a.ldH = p.ldD; a.kabs0 = p.kcut0;
    a.Wpart = p.get<cplx>("Wpart"); a.W = p.get<cplx>("W"); a.nWG = p.nWG; a.dpw = 0;
    const int k0 = std::max(p.kcut0, 1);
    a.kfirst = k0;
    p.sweep_launches = 0;
    for (int kb = k0; kb < p.P; ++kb) { launch_sweep_wide(a, kb, true, st); ++p.sweep_launches; }
    if (k0 < p.P) launch_sweep_wide_finalize(p.get("Wpart"), p.get("W"), p.nWG, M, p.P, p.P - 1, st);
    p.mark("magls_sweep");
    launch_filter_epilogue(p.get("W"), M, p.nfft, (int)d.len, p.get("tw"), p.get<double>("grpd"), 0, 1, 1, 0, p.get("wL"), p.get("wR"), st);
    p.mark("epilogue");
}

void execute_from_atf(emagls_plan& p) {
    if (p.wide) { execute_from_atf_wide(p); return; }
    // (eager / profiled executes; plan_execute captures the stages around the sweep otherwise.  More matched directions than the
    // dense route's QR holds: the Gram route as well, emagls_run_sweep then launches bin by bin)
    if (p.sweep_persist || p.Dm > 4096) {
        from_atf_pre_sweep(p);
        emagls_run_sweep(p);
        from_atf_post_sweep(p);
        return;
    }
    hipStream_t st = p.stream;
    const emagls_design_desc& d = p.d;
    const int M = p.C;
    // ---- grid matching (FromAtf.m:56-95)
    if (p.hrir_smaller)
        launch_grid_match(p.get<double>("hrir_azi"), p.get<double>("hrir_zen"), d.ndirs, p.get<double>("atf_azi"),
                          p.get<double>("atf_zen"), d.natf, p.get<double>("cartB"), p.get<int64_t>("match_idx"),
                          p.get<double>("match_dev"), p.get<double>("mean_dev"), st);
    else
        launch_grid_match(p.get<double>("atf_azi"), p.get<double>("atf_zen"), d.natf, p.get<double>("hrir_azi"),
                          p.get<double>("hrir_zen"), d.ndirs, p.get<double>("cartB"), p.get<int64_t>("match_idx"),
                          p.get<double>("match_dev"), p.get<double>("mean_dev"), st);
    launch_atf_colidx(p.hrir_smaller ? p.get<int64_t>("match_idx") : nullptr, p.Dm, M, p.get<int64_t>("colidx"), st);
    p.mark("grid_match");
    stage_prologue(p, 1, p.hrir_smaller ? nullptr : p.get<int64_t>("match_idx"), p.Dm);
    // atfs = fft(atfIrs, nfft) on the matched directions only: X[kb][m][d]
    launch_real_fft_gather(p.get<double>("atf"), d.atf_taps, (int64_t)M * p.Dm, p.get<int64_t>("colidx"), p.nfft, p.get("tw"),
                           p.get("X"), (int64_t)M * p.ldD, p.Dm, p.ldD, st);
    p.mark("atf_fft");
    const int ls_end = std::min(p.kcut0, p.P);
    {
        FactorArgs a{};
        a.S = (int)p.Dm; a.C = M; a.ldS = (int)p.ldD; a.kb0 = 1; a.P = p.P;
        a.Xd = p.get<cplx>("X"); a.xd_stride = (int64_t)M * p.ldD;
        a.reg_mode = 0; a.reg_c = SVD_REGUL_CONST;
        a.Z = p.get<cplx>("Z"); a.Vws = p.get<cplx>("Vws"); a.sv = p.get<double>("sv");
        a.Hq = p.get<cplx>("Hc"); a.ldHq = p.ldD; a.hq_estride = (int64_t)ls_end * p.ldD; a.ls_end = ls_end;
        a.W = p.get<cplx>("W"); a.sweeps_out = p.get<int>("jsweeps");
        a.tauw = p.get<double>("tauw"); a.R2w = p.get<cplx>("R2w"); a.Nw = p.get<cplx>("Nw");
        launch_factor(a, p.P - 1, true, st);
    }
    p.mark("factor_bins");
    {
        DenseSweepArgs a{};
        a.D = (int)p.Dm; a.C = M; a.ldD = (int)p.ldD; a.P = p.P;
        a.X = p.get("X"); a.x_stride = (int64_t)M * p.ldD; a.Zd = p.get("Z"); a.z_stride = (int64_t)M * p.ldD;
        a.Habs = p.get<double>("Habs"); a.ldH = p.ldD; a.kabs0 = p.kcut0;
        a.Wpart = p.get<cplx>("Wpart"); a.W = p.get<cplx>("W"); a.nWG = p.nWG;
        const int k0 = std::max(p.kcut0, 1);
        a.kfirst = k0;
        p.sweep_launches = 0;
        for (int kb = k0; kb < p.P; ++kb) {
            if (p.prof_level >= 2) record_sweep_event(p, 2 * (size_t)p.sweep_launches);
            launch_sweep_dense(a, kb, true, st);
            if (p.prof_level >= 2) record_sweep_event(p, 2 * (size_t)p.sweep_launches + 1);
            ++p.sweep_launches;
        }
        if (k0 < p.P) launch_sweep_finalize(p.get("Wpart"), p.get("W"), p.nWG, M, p.P, p.P - 1, st);
    }
    p.mark("magls_sweep");
    launch_filter_epilogue(p.get("W"), M, p.nfft, (int)d.len, p.get("tw"), p.get<double>("grpd"), 0, 1, 1, 0, p.get("wL"),
                           p.get("wR"), st);
    p.mark("epilogue");
}

void run_pipeline(emagls_plan& p) {
    const emagls_design_desc& d = p.d;
    p.stage_names.clear();
    launch_zero(p.get("flag"), sizeof(int) * NFLAG, p.stream);
    if (p.has("route")) launch_zero(p.get("route"), p.bufs["route"].bytes, p.stream);
    if (p.has("W")) launch_zero(p.get("W"), p.bufs["W"].bytes, p.stream);
    p.mark("begin");
    switch (d.kind) {
        case EMAGLS_KIND_LS: execute_ls(p); break;
        case EMAGLS_KIND_MAGLS:
        case EMAGLS_KIND_MAGLS_2D: execute_magls(p); break;
        case EMAGLS_KIND_EMAGLS:
        case EMAGLS_KIND_EMAGLS2:
        case EMAGLS_KIND_EMA_CH:
        case EMAGLS_KIND_EMA_SH: execute_emagls(p); break;
        default: execute_from_atf(p); break;
    }
}

void plan_pre_stage(emagls_plan& p);
template <typename F> void capture_into(hipStream_t st, hipGraph_t* g, hipGraphExec_t* ge, F&& body) {
    HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    try {
        body();
    } catch (...) {
        hipGraph_t tmp = nullptr;
        hipStreamEndCapture(st, &tmp);
        if (tmp) hipGraphDestroy(tmp);
        throw;
    }
    HIP_CHECK(hipStreamEndCapture(st, g));
    HIP_CHECK(hipGraphInstantiate(ge, *g, nullptr, nullptr, 0));
}

void plan_execute(emagls_plan& p) {
    const emagls_design_desc& d = p.d;
    if (p.custom_basis) {
        if (!p.have_basis || !p.have_hrirs) throw Error(EMAGLS_ERR_ARG, "HRIRs and the SH matrices must be set before execute");
    } else {
        if (!p.have_hrir_grid || !p.have_hrirs) throw Error(EMAGLS_ERR_ARG, "HRIRs and their grid must be set before execute");
        if (array_kind(d.kind) && !p.have_mic_grid)
            throw Error(EMAGLS_ERR_ARG, "microphone grid must be set before execute");
    }
    if (d.kind == EMAGLS_KIND_FROM_ATF && !p.have_atfs) throw Error(EMAGLS_ERR_ARG, "ATFs must be set before execute");
    if (p.nstreams >= 2) p.need_sides(p.nstreams);   // (before any capture begins)
    const bool persist = d.kind != EMAGLS_KIND_LS && p.sweep_persist;
    if (p.prof_level == 0 && p.use_graph && persist) {
        // the persistent sweep is launched directly (SweepChain); the stages before it are captured from the second
        // execute on (the first runs eagerly: one-time function attributes, lazy module load)
        // a design with forked stages (it has the device to itself) runs what its sweep does not need -- Cholesky factor, orthonormal
        // route of the low bins: plan_defers_hh_route -- NEXT to the sweep, eagerly on a stream of its own (a dozen launches)
        if (!p.pre_exec) p.defer_hh = (p.nstreams >= 2 || p.alone) && array_kind(d.kind) && plan_defers_hh_route(p);
        p.pre_phase = p.defer_hh ? 1 : 0;
        try {
            if (!p.pre_exec && p.eager_runs >= 1) capture_into(p.stream, &p.pre_graph, &p.pre_exec, [&] { plan_pre_stage(p); });
            if (p.pre_exec) HIP_CHECK(hipGraphLaunch(p.pre_exec, p.stream)); else plan_pre_stage(p);
        } catch (...) { p.pre_phase = 0; throw; }
        p.pre_phase = 0;
        if (p.defer_hh) {
            if (!p.hh_stream) p.hh_stream = StreamPool::get().take();
            p.depend(p.hh_stream, p.stream);   // (behind the stages the sweep needs, before the sweep is enqueued)
        }
        emagls_run_sweep(p);
        if (p.defer_hh) {
            hipStream_t keep = p.stream;
            const int keep_n = p.nstreams;
            p.stream = p.hh_stream; p.nstreams = 1; p.pre_phase = 2;
            try { emagls_pre_sweep(p); } catch (...) { p.stream = keep; p.nstreams = keep_n; p.pre_phase = 0; throw; }
            p.stream = keep; p.nstreams = keep_n; p.pre_phase = 0;
            p.depend(p.stream, p.hh_stream);   // (the epilogue reads the rows of every bin)
        }
        if (d.kind == EMAGLS_KIND_FROM_ATF) from_atf_post_sweep(p);
        else if (magls_kind(d.kind)) magls_post_sweep(p);
        else emagls_post_sweep(p);
        if (!p.pre_exec) ++p.eager_runs;
        p.executed = true;
        return;
    }
    // sets of one geometry through a plan of the 33..64-channel path: the stages that depend on the grids alone are kept from the last clean run
    // on these grids; such an execute runs eagerly (a thousand launches of 12 us each: the host stays ahead)
    if (p.geo_keep && p.wide && p.prof_level == 0 && (d.kind == EMAGLS_KIND_EMAGLS || d.kind == EMAGLS_KIND_EMAGLS2) &&
        p.geo_done_version == p.atf_side_version) {
        p.geo_skip = true;
        try { run_pipeline(p); } catch (...) { p.geo_skip = false; throw; }
        p.geo_skip = false;
        p.executed = true;
        return;
    }
    p.geo_run_version = p.atf_side_version;
    if (p.prof_level == 0 && p.use_graph) {
        // first execute runs eagerly (one-time function attributes, lazy module load), the second is captured
        if (!p.graph_exec && p.eager_runs >= 1) {
            HIP_CHECK(hipStreamBeginCapture(p.stream, hipStreamCaptureModeThreadLocal));
            try {
                run_pipeline(p);
            } catch (...) {
                hipGraph_t g = nullptr;
                hipStreamEndCapture(p.stream, &g);
                if (g) hipGraphDestroy(g);
                throw;
            }
            HIP_CHECK(hipStreamEndCapture(p.stream, &p.graph));
            HIP_CHECK(hipGraphInstantiate(&p.graph_exec, p.graph, nullptr, nullptr, 0));
        }
        if (p.graph_exec) {
            HIP_CHECK(hipGraphLaunch(p.graph_exec, p.stream));
            p.executed = true;
            return;
        }
    }
    run_pipeline(p);
    ++p.eager_runs;
    p.executed = true;
}

void emagls_pre_sweep(emagls_plan& p);
void emagls_post_sweep(emagls_plan& p);
// A batch runs as separate graphs on separate streams (one hipGraph executes its nodes in order, so
// parallel branches inside ONE graph would serialize): per-plan "pre" graphs on the plans' own streams,
// the shared sweep graph on the batch stream, ordered by events outside the graphs.
void batch_execute_lanes(emagls_batch& b);
void plan_pre_stage(emagls_plan& p) {
    p.stage_names.clear();
    launch_zero(p.get("flag"), sizeof(int) * NFLAG, p.stream);
    if (p.has("route")) launch_zero(p.get("route"), p.bufs["route"].bytes, p.stream);
    launch_zero(p.get("W"), p.bufs["W"].bytes, p.stream);
    if (p.d.kind == EMAGLS_KIND_FROM_ATF) from_atf_pre_sweep(p);
    else if (magls_kind(p.d.kind)) magls_pre_sweep(p);
    else emagls_pre_sweep(p);
}
void batch_sweep_stage(emagls_batch& b) {
    const int nb = (int)b.plans.size();
    std::vector<HalfSweepArgs> ha((size_t)nb);
    for (int j = 0; j < nb; ++j) ha[j] = emagls_half_args(*b.plans[j]);
    if (b.atf_share || b.geo_share)   // one ATF side / one geometry for every subject
        for (int j = 1; j < nb; ++j) {
            ha[j].G = ha[0].G; ha[j].Yri = ha[0].Yri; ha[j].Mw = ha[0].Mw; ha[j].cond_ok = ha[0].cond_ok;
            ha[j].bsc = ha[0].bsc; ha[j].smap = ha[0].smap;   // (synthesising sweep: plan 0's scaled modal terms and Mt; the grids are the same by the sharing check)
            ha[j].skip_flag = ha[0].skip_flag;   // (MagLS: plan 0 judged the basis for everybody)
        }
    const bool reg = b.plans[0]->sweep_persist && reg_sweep_wanted(b.plans.data(), nb);
    for (auto* q : b.plans) q->reg_sweep = reg;
    if (!reg && nb > SWEEP_MULTI_MAX) throw Error(EMAGLS_ERR_UNSUPPORTED, "internal: more than 16 designs in a batch need the register-resident sweep");
    HalfSweepMulti h{};
    h.n = std::min(nb, SWEEP_MULTI_MAX);
    for (int j = 0; j < h.n; ++j) h.a[j] = ha[j];
    emagls_plan& q0 = *b.plans[0];
    const int kk0 = std::max(q0.kcut0, 1);
    if (kk0 >= q0.P) return;
    if (q0.sweep_persist) {
        // (tried in round 4: the launch on a stream of the highest priority, so that the dispatcher places the sweep's workgroups
        // before other batches' refilling kernels -- per batch, or one per device: 1744 against 1646 sets/s at 20 steps in one
        // session, nothing in the next, and with six / eight batches in flight the extra streams, multiplexed onto hardware
        // queues that held each other's waits, stalled runs for seconds (28-880 sets/s): rejected)
        hipStream_t ss = b.stream;
        {
            SweepGate gate(ss, reg ? reg_sweep_gate_cost((int)q0.D, nb) : 0);
            if (b.lanes) { BatchScope sc(nb, b.stride); launch_zero(q0.get("ll"), q0.bufs["ll"].bytes, ss); }   // (one launch for every lane)
            else for (auto* q : b.plans) launch_zero(q->get("ll"), q->bufs["ll"].bytes, ss);
            if (reg) {
                if (!b.sweep_args_dev) HIP_CHECK(hipMalloc(&b.sweep_args_dev, sizeof(HalfSweepArgs) * (size_t)REG_SWEEP_MAX));
                reg_args_upload(ha.data(), nb, b.sweep_args_dev, b.sweep_args_last, ss);
            }
            if (b.prof_level >= 1) HIP_CHECK(hipEventRecord(b.sweep_ev[0], ss));
            if (reg) launch_sweep_reg(static_cast<const HalfSweepArgs*>(b.sweep_args_dev), ha[0], nb, ss);
            else if (q0.synth) launch_sweep_synth(h, ss); else launch_sweep_persist(h, ss);
            if (b.prof_level >= 1) HIP_CHECK(hipEventRecord(b.sweep_ev[1], ss));
        }
        if (ss != b.stream) b.depend(b.stream, ss);
        return;
    }
    for (int kb = kk0; kb < q0.P; ++kb) launch_sweep_half(h, kb, b.stream);
    launch_sweep_half_finalize(h, q0.P - 1, b.stream);
}

// lane mode: the pipeline of plan `first` is enqueued once on `st` with grid.z = `count` designs (plans first .. first + count - 1)
// part 0: stages before the sweep, part 2: stages after it
// EMAGLS_STAGGER (default 1): the two lane groups of a batch issue the stages before the sweep in complementary orders
// (emagls_pre_sweep, orders 1 and 2); 0: both in the order of a single design
static int stagger_mode() {
    static const int m = [] { const char* e = getenv("EMAGLS_STAGGER"); return e ? atoi(e) : 1; }();
    return m;
}
// a lane batch whose designs all keep the orthonormal route of their low bins off the path to the sweep (plan_defers_hh_route)
bool batch_defers_hh(const emagls_batch& b) {
    // Only a batch that has the device to itself (a job list of ONE chunk: b.alone, set with its forked streams): there the stages are
    // the path to the sweep -- 2570-2620 -> 2810-2880 sets/s at 20 steps although the sweep itself runs 10 % longer next to them.  With
    // four chunks in flight the same work only moves, and the slower sweeps cost 6 % (3470 -> 3270 at 128 steps).
    static const int defer_mode = [] { const char* e = getenv("EMAGLS_DEFER_HH"); return e ? atoi(e) : 1; }();   // (2: every lane batch -- experiments)
    if (!b.lanes || !(b.alone || defer_mode == 2) || b.geo_share || b.atf_share) return false;
    for (const emagls_plan* p : b.plans) if (!p || !plan_defers_hh_route(*p)) return false;
    return true;
}
void batch_lanes_part(emagls_batch& b, int part, int first, int count, hipStream_t st, int group = 0) {
    emagls_plan& p0 = *b.plans[first];
    hipStream_t keep = p0.stream, keep_side[3] = {p0.side[0], p0.side[1], p0.side[2]};
    const int keep_streams = p0.nstreams, keep_order = p0.stage_order;
    p0.stream = st;
    p0.nstreams = (part == 0 && b.groups == 1) ? b.nstreams : 1;
    p0.stage_order = 0;
    if (part == 0 && p0.nstreams == 1) {
        const int sm = stagger_mode();
        if (sm == 1) p0.stage_order = b.groups > 1 ? 1 + group % 2 : (b.order_hint ? 1 + (b.order_hint - 1) % 2 : 0);
        else if (sm >= 10) p0.stage_order = group == 0 ? sm / 10 % 10 : sm % 10;   // (experiments: "12", "21", "11", "22")
    }
    if (p0.nstreams > 1) for (int i = 0; i < 3; ++i) p0.side[i] = b.side[i];
    const int keep_phase = p0.pre_phase;
    // part 0 of a batch that runs the orthonormal route next to its sweep: only what the sweep needs; part 3: the rest
    p0.pre_phase = part == 3 ? 2 : (part == 0 && b.defer_hh) ? 1 : 0;
    auto restore = [&] { p0.stream = keep; p0.nstreams = keep_streams; p0.stage_order = keep_order; p0.pre_phase = keep_phase; for (int i = 0; i < 3; ++i) p0.side[i] = keep_side[i]; };
    try {
        BatchScope sc(count, b.stride);
        if (part == 0) plan_pre_stage(p0); else if (part == 3) emagls_pre_sweep(p0); else emagls_post_sweep(p0);
    } catch (...) {
        restore();
        throw;
    }
    restore();
}
// Lane GROUPS: a batch of more than 8 designs runs the stages before its sweep as two half-batches on two streams (each one
// launch of every kernel for its lanes, each a captured single-stream graph) and then ONE resident sweep launch for all designs.
// Sixteen lanes in one launch sequence take about twice as long per kernel as eight (the bandwidth-bound kernels scale with
// the lanes, and the latency-bound ones get 2x the workgroups), and that sequence is the path to the sweep; two half-batches
// next to each other overlap their latency-bound kernels (measured at --steps 20: the 16-lane sequence 8.6 ms).
int batch_group_first(const emagls_batch& b, int g) { const int n = (int)b.plans.size(), h = (n + b.groups - 1) / b.groups; return std::min(n, g * h); }
void batch_execute_lanes(emagls_batch& b) {
    const bool replay = b.use_graph && b.eager_runs >= 1;
    const int n = (int)b.plans.size();
    const int ng = b.groups;
    for (int g = 1; g < ng; ++g) if (!b.side[g - 1]) b.side[g - 1] = emagls::pool_stream_take();
    hipStream_t gs[4] = {b.stream, b.stream, b.stream, b.stream};
    for (int g = 1; g < ng; ++g) gs[g] = b.side[g - 1];
    hipGraph_t* gr[4] = {&b.graph, &b.graph2, &b.graphx[0], &b.graphx[1]};
    hipGraphExec_t* ge[4] = {&b.graph_exec, &b.graph2_exec, &b.graphx_exec[0], &b.graphx_exec[1]};
    // the stages the sweep does not need (Cholesky factor, orthonormal route of the low bins) run NEXT to it, one stream per lane group
    // (decided on the eager run and when the graphs are captured; a replay keeps what its graphs were captured with)
    if (!replay || !b.graph_exec) b.defer_hh = batch_defers_hh(b);
    const bool defer = b.defer_hh;
    if (defer) for (int g = 0; g < ng; ++g) if (!b.hh_stream[g]) b.hh_stream[g] = emagls::pool_stream_take();
    if (replay && !b.graph_exec) {
        for (int g = 0; g < ng; ++g) {
            const int f = batch_group_first(b, g), c = batch_group_first(b, g + 1) - f;
            capture_into(gs[g], gr[g], ge[g], [&] { batch_lanes_part(b, 0, f, c, gs[g], g); });
            if (defer) capture_into(b.hh_stream[g], &b.graph_hh[g], &b.graph_hh_exec[g], [&] { batch_lanes_part(b, 3, f, c, b.hh_stream[g], g); });
        }
        capture_into(b.stream, &b.post_graph, &b.post_exec, [&] { batch_lanes_part(b, 2, 0, n, b.stream); });
    }
    b.used = 0;
    for (int g = 1; g < ng; ++g) b.depend(gs[g], b.stream);   // (the previous execute of this batch is done with the buffers)
    if (replay && ng > 1) {
        // a graph launch of ~60 kernel nodes costs ~1 ms of host time: the other groups' launches go out from threads of their
        // own, or their stages would start a millisecond (three under a profiler) behind the first group's
        hipError_t err[4] = {hipSuccess, hipSuccess, hipSuccess, hipSuccess};
        const int dev = b.device;
        std::vector<std::thread> th;
        for (int g = 1; g < ng; ++g)
            th.emplace_back([&, g] { err[g] = hipSetDevice(dev); if (err[g] == hipSuccess) err[g] = hipGraphLaunch(*ge[g], gs[g]); });
        err[0] = hipGraphLaunch(*ge[0], gs[0]);
        for (auto& t : th) t.join();
        for (int g = 0; g < ng; ++g) HIP_CHECK(err[g]);
    } else {
        for (int g = 0; g < ng; ++g) {
            const int f = batch_group_first(b, g), c = batch_group_first(b, g + 1) - f;
            if (replay) HIP_CHECK(hipGraphLaunch(*ge[g], gs[g])); else batch_lanes_part(b, 0, f, c, gs[g], g);
        }
    }
    if (defer) for (int g = 0; g < ng; ++g) b.depend(b.hh_stream[g], gs[g]);   // (behind the group's stages, before the sweep is enqueued)
    for (int g = 1; g < ng; ++g) b.depend(b.stream, gs[g]);
    batch_sweep_stage(b);   // (never captured: see SweepGate)
    if (defer) {
        for (int g = 0; g < ng; ++g) {
            const int f = batch_group_first(b, g), c = batch_group_first(b, g + 1) - f;
            if (replay) HIP_CHECK(hipGraphLaunch(b.graph_hh_exec[g], b.hh_stream[g])); else batch_lanes_part(b, 3, f, c, b.hh_stream[g], g);
            b.depend(b.stream, b.hh_stream[g]);   // (the epilogue reads the rows of every bin)
        }
    }
    if (replay) HIP_CHECK(hipGraphLaunch(b.post_exec, b.stream)); else batch_lanes_part(b, 2, 0, n, b.stream);
    emagls_plan& p0 = *b.plans[0];
    for (auto* p : b.plans) {
        p->executed = true;
        p->sweep_launches = p0.sweep_persist ? 1 : p0.P - std::max(p0.kcut0, 1);
    }
    if (!replay) ++b.eager_runs;
}

void drop_plan_graphs(emagls_plan& p);
void drop_batch_graphs(emagls_batch& b);
// FromAtf subjects share their ATF side when every plan holds the same grids and ATF set and no bin needs the dense route
void batch_atf_decide_sharing(emagls_batch& b) {
    uint64_t ver = 0;
    for (auto* p : b.plans) ver = ver * 1000003ull + p->atf_side_version;
    bool same = true;
    if (ver != b.atf_checked_version) {
        if (!b.cmp_flag) HIP_CHECK(hipMalloc(&b.cmp_flag, 16));
        HIP_CHECK(hipStreamSynchronize(b.stream));
        HIP_CHECK(hipMemsetAsync(b.cmp_flag, 0, 16, b.stream));
        emagls_plan& p0 = *b.plans[0];
        for (size_t j = 1; j < b.plans.size(); ++j)
            for (const char* name : {"atf", "atf_azi", "atf_zen", "hrir_azi", "hrir_zen"})
                launch_compare_words(p0.get(name), b.plans[j]->get(name), p0.bufs[name].bytes, b.cmp_flag, b.stream);
        int differ = 0;
        HIP_CHECK(hipMemcpyAsync(&differ, b.cmp_flag, sizeof differ, hipMemcpyDeviceToHost, b.stream));
        HIP_CHECK(hipStreamSynchronize(b.stream));
        b.atf_checked_version = ver;
        same = differ == 0;
    } else {
        same = b.atf_inputs_same;   // (nothing was replaced since the last comparison)
    }
    b.atf_inputs_same = same;
    bool routes_ok = true;
    for (auto* p : b.plans) routes_ok = routes_ok && p->gram_from == 1 && p->sweep_persist == b.plans[0]->sweep_persist;
    const bool share = same && routes_ok && b.plans.size() > 1;
    if (share != b.atf_share) {   // the captured per-plan stages differ between the two modes
        for (auto* p : b.plans) drop_plan_graphs(*p);
        drop_batch_graphs(b);
        b.atf_share = share;
    }
}
void from_atf_subject_pre_stage(emagls_plan& p) {   // a subject of a sharing batch: everything but the ATF side
    p.stage_names.clear();
    launch_zero(p.get("flag"), sizeof(int) * NFLAG, p.stream);
    launch_zero(p.get("W"), p.bufs["W"].bytes, p.stream);
    from_atf_subject_stage(p);
}
// Subjects of ONE ATF set on ONE HRIR grid (checked on the device): the whole batch as two single-stream graphs around the
// resident sweep -- plan 0's full stage (grid match, ATF spectra, per-bin factors), then per subject only the HRIR prologue
// (plan 0's grid match serves every subject) and the least-squares rows; afterwards every subject's epilogue.  Eight graphs
// on eight streams with an event pair each (the earlier form) cost more host time than the stages take on the GPU: 10.2 ms
// per batch of BASELINE config 5, of which 3.3 ms are the sweep and ~4 ms kernels that could overlap.
void batch_atf_shared_stage(emagls_batch& b, int part) {
    emagls_plan& p0 = *b.plans[0];
    std::vector<hipStream_t> keep;
    for (auto* p : b.plans) { keep.push_back(p->stream); p->stream = b.stream; }
    auto restore = [&] { for (size_t j = 0; j < b.plans.size(); ++j) b.plans[j]->stream = keep[j]; };
    try {
        if (part == 0) {
            plan_pre_stage(p0);
            for (size_t j = 1; j < b.plans.size(); ++j) {
                emagls_plan& p = *b.plans[j];
                p.stage_names.clear();
                launch_zero(p.get("flag"), sizeof(int) * NFLAG, b.stream);
                launch_zero(p.get("W"), p.bufs["W"].bytes, b.stream);
                // (the subject keeps its own copy of the match: emagls_plan_get_info and the debug buffers read it per plan)
                // (plain device-to-device copies: match_idx holds 64-bit integers)
                HIP_CHECK(hipMemcpyAsync(p.get("match_idx"), p0.get("match_idx"), sizeof(int64_t) * (size_t)p.Dm, hipMemcpyDeviceToDevice, b.stream));
                HIP_CHECK(hipMemcpyAsync(p.get("match_dev"), p0.get("match_dev"), sizeof(double) * (size_t)p.Dm, hipMemcpyDeviceToDevice, b.stream));
                HIP_CHECK(hipMemcpyAsync(p.get("mean_dev"), p0.get("mean_dev"), sizeof(double), hipMemcpyDeviceToDevice, b.stream));
                stage_prologue(p, 1, p.hrir_smaller ? nullptr : p.get<int64_t>("match_idx"), p.Dm);
                from_atf_ls_rows(p, p0, b.stream);
            }
        } else {
            for (auto* p : b.plans) from_atf_post_sweep(*p);
        }
    } catch (...) { restore(); throw; }
    restore();
}
void batch_execute_atf(emagls_batch& b) {
    for (auto* p : b.plans)
        if (!p->have_hrirs || !p->have_hrir_grid || !p->have_atfs)
            throw Error(EMAGLS_ERR_ARG, "every plan of the batch needs its HRIRs, its grid and the ATFs");
    batch_atf_decide_sharing(b);
    const bool replay = b.use_graph && b.eager_runs >= 1;
    emagls_plan& p0 = *b.plans[0];
    static const bool one_stream = [] { const char* e = getenv("EMAGLS_ATF_ONE_STREAM"); return !(e && e[0] == '0'); }();
    if (b.atf_share && p0.sweep_persist && one_stream) {
        if (replay && !b.graph_exec) {
            capture_into(b.stream, &b.graph, &b.graph_exec, [&] { batch_atf_shared_stage(b, 0); });
            capture_into(b.stream, &b.post_graph, &b.post_exec, [&] { batch_atf_shared_stage(b, 2); });
        }
        b.used = 0;
        if (replay) HIP_CHECK(hipGraphLaunch(b.graph_exec, b.stream)); else batch_atf_shared_stage(b, 0);
        batch_sweep_stage(b);   // (never captured: see SweepChain)
        if (replay) HIP_CHECK(hipGraphLaunch(b.post_exec, b.stream)); else batch_atf_shared_stage(b, 2);
        for (auto* p : b.plans) { p->executed = true; p->sweep_launches = 1; }
        if (!replay) ++b.eager_runs;
        return;
    }
    auto pre = [&](emagls_plan& p) { if (b.atf_share && &p != &p0) from_atf_subject_pre_stage(p); else plan_pre_stage(p); };
    if (replay && !p0.pre_exec) {
        for (auto* p : b.plans) capture_into(p->stream, &p->pre_graph, &p->pre_exec, [&] { pre(*p); });
        if (!p0.sweep_persist) capture_into(b.stream, &b.graph, &b.graph_exec, [&] { batch_sweep_stage(b); });
    }
    b.used = 0;
    for (auto* p : b.plans) b.depend(p->stream, b.stream);   // (the previous execute of this batch is done with the buffers)
    for (auto* p : b.plans) {
        if (replay) HIP_CHECK(hipGraphLaunch(p->pre_exec, p->stream)); else pre(*p);
        b.depend(b.stream, p->stream);
    }
    if (b.atf_share)   // least-squares bins of the other subjects on plan 0's operands
        for (size_t j = 1; j < b.plans.size(); ++j) from_atf_ls_rows(*b.plans[j], p0, b.stream);
    if (p0.sweep_persist) batch_sweep_stage(b);   // (never captured: see SweepChain)
    else if (replay) HIP_CHECK(hipGraphLaunch(b.graph_exec, b.stream)); else batch_sweep_stage(b);
    for (auto* p : b.plans) {
        b.depend(p->stream, b.stream);
        from_atf_post_sweep(*p);
        b.depend(b.stream, p->stream);  // batch stream completion == all results ready
        p->executed = true;
        p->sweep_launches = p0.sweep_persist ? 1 : p0.P - std::max(p0.kcut0, 1);
    }
    if (!replay) ++b.eager_runs;
}

// ---------------------------------------------------------------------------------------------
// Batches of HRIR sets on ONE geometry (north_star: independent jobs "per HRTF set"): same HRIR grid, same array, same
// orders and lengths.  Everything of lib/getEMagLsFilters.m:44-70,85-93 -- the SH matrices, the array model, pwGrid_k and its
// regularised inverse of every bin -- depends on the geometry only; the HRIR set enters through the spectra (:72-81), the
// least-squares rows (:94) and the sweep's magnitudes (:99-102).  Plan 0 runs the whole pipeline; the other plans run their
// HRIR prologue, their least-squares rows on plan 0's factors, and sweep on plan 0's G_k / M_k (batch_sweep_stage).
// ---------------------------------------------------------------------------------------------
void batch_geo_decide_sharing(emagls_batch& b) {
    bool share = false;
    if (b.geo_want && b.plans.size() > 1) {
        emagls_plan& p0 = *b.plans[0];
        bool eligible = (p0.d.kind == EMAGLS_KIND_EMAGLS || p0.d.kind == EMAGLS_KIND_EMAGLS2 || p0.d.kind == EMAGLS_KIND_EMA_CH) && !p0.wide &&
                        !p0.diffuse && !p0.custom_basis;
        for (auto* p : b.plans) {
            const emagls_design_desc &x = p->d, &y = p0.d;
            eligible = eligible && x.kind == y.kind && x.order == y.order && x.fs == y.fs && x.len == y.len && x.nsamp == y.nsamp &&
                       x.ndirs == y.ndirs && x.mic_radius == y.mic_radius && x.nmics == y.nmics && x.basis == y.basis &&
                       x.sim_order_pad == y.sim_order_pad && p->wide == p0.wide && p->diffuse == p0.diffuse && p->custom_basis == p0.custom_basis &&
                       p->real_internal == p0.real_internal && p->gram_from == p0.gram_from && p->hh_end == p0.hh_end && p->n_h == p0.n_h &&
                       p->g0 == p0.g0 && p->sweep_persist == p0.sweep_persist;
        }
        if (eligible) {
            uint64_t ver = 0;
            for (auto* p : b.plans) ver = ver * 1000003ull + p->atf_side_version;
            if (ver != b.geo_checked_version) {
                if (!b.cmp_flag) HIP_CHECK(hipMalloc(&b.cmp_flag, 16));
                HIP_CHECK(hipStreamSynchronize(b.stream));
                HIP_CHECK(hipMemsetAsync(b.cmp_flag, 0, 16, b.stream));
                for (size_t j = 1; j < b.plans.size(); ++j)
                    for (const char* name : {"hrir_azi", "hrir_zen", "mic_azi", "mic_zen"})
                        launch_compare_words(p0.get(name), b.plans[j]->get(name), p0.bufs[name].bytes, b.cmp_flag, b.stream);
                int differ = 0;
                HIP_CHECK(hipMemcpyAsync(&differ, b.cmp_flag, sizeof differ, hipMemcpyDeviceToHost, b.stream));
                HIP_CHECK(hipStreamSynchronize(b.stream));
                b.geo_checked_version = ver;
                b.geo_inputs_same = differ == 0;
            }
            share = b.geo_inputs_same;
        }
    }
    if (share != b.geo_share) {   // (the two modes enqueue different stages: nothing captured for the other one may be replayed)
        for (auto* p : b.plans) drop_plan_graphs(*p);
        drop_batch_graphs(b);
        b.geo_share = share;
    }
}
// a subject of a geometry-sharing batch, first part: what needs its HRIRs only (lib/getEMagLsFilters.m:72-81)
void emagls_subject_prologue(emagls_plan& p, const emagls_plan& g) {
    const emagls_design_desc& d = p.d;
    hipStream_t st = p.stream;
    const int ls_end = std::min(g.kcut0, g.P);
    p.stage_names.clear();
    p.sync_used = 0;
    launch_zero(p.get("flag"), sizeof(int) * NFLAG, st);
    launch_zero(p.get("W"), p.bufs["W"].bytes, st);
    launch_twiddles(p.nfft, p.get("tw"), st);
    launch_hrir_grpdelay(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, d.ndirs, p.nfft, p.get("tw"), p.get<double>("dirsum"),
                         p.get<double>("grpd"), st);
    launch_hrir_fft(p.get<double>("hL"), p.get<double>("hR"), d.nsamp, p.D, nullptr, p.nfft, p.get("tw"), p.get<double>("grpd"), 0, ls_end,
                    p.kcut0, p.get("Hc"), p.get<double>("Habs"), p.ldD, st, ls_end > 0 ? p.get<double>("HcT") : nullptr,
                    round_up(4 * std::max(ls_end, 1), 64));
}
// second part, behind plan g's stages: the least-squares bins (:94) on g's factors -- H conj(Q) R^-1 rows and the
// back-transform of the Householder-route bins (into the subject's own Z), G_k / M_k of g for the Gram-route bins
void emagls_subject_rows(emagls_plan& p, emagls_plan& g) {
    hipStream_t st = p.stream;
    const bool cb = g.cplx_basis;
    const int gf = g.gram_from, hh_end = g.hh_end, Sh = g.S_h, ldSh = g.ldS_h, nOrdH = g.n_h + 1, nOrd = g.simOrder + 1;
    const int ls_end = std::min(g.kcut0, g.P);
    const int ls_h = std::min(ls_end, hh_end);
    const int64_t g_stride = (int64_t)g.C * g.ldD;
    if (hh_end > 1) {
        launch_hy_conj_mfma(p.get<double>("HcT"), round_up(4 * std::max(ls_end, 1), 64), ls_end, g.get("Yc"), g.ldS, cb, (int)g.D, Sh,
                            p.get<double>("Hyp"), p.get("Hq"), ldSh, st);
        launch_qform(p.get("Hq"), g.get(cb ? "R" : "Rc"), p.get(cb ? "Rinv" : "Rinvc"), Sh, 2 * (int64_t)std::max(ls_end, 1), ldSh, true, p.get("Hq"), st);
        FactorArgs fa{};
        fa.S = Sh; fa.C = g.C; fa.ldS = ldSh; fa.kb0 = 1; fa.P = g.P;
        fa.Tn = g.get("Tn"); fa.bn = g.get<cplx>("bn"); fa.nOrders = nOrdH; fa.bn_stride = nOrd;
        fa.reg_mode = 0; fa.reg_c = SVD_REGUL_CONST;
        fa.Z = p.get<cplx>("Z");
        fa.Mw = g.get<cplx>("Mw");
        fa.Vws = g.get<cplx>("Vws"); fa.sv = g.get<double>("sv");
        fa.Hq = p.get<cplx>("Hq"); fa.ldHq = ldSh; fa.hq_estride = (int64_t)ls_end * ldSh; fa.ls_end = ls_h;
        fa.hq_conj = 1;
        fa.route = g.get<int>("route"); fa.status = p.get<int>("flag");
        fa.cond_limit = 10.0 * GRAM_COND_EST;
        fa.W = p.get<cplx>("W"); fa.sweeps_out = nullptr;
        fa.tauw = g.get<double>("tauw"); fa.R2w = g.get<cplx>("R2w"); fa.Nw = g.get<cplx>("Nw");
        fa.cond_ok = g.get<double>("cond_ok");
        launch_factor(fa, hh_end - 1, cb, st, 2);
    }
    if (gf > 0 && gf < ls_end) {
        if (g.synth) {   // (lane launches: the subject's own copies of the grids and of the row order, plan 0's coefficients, Pm and M_k)
            const emagls_plan& g0p = g.geo_from ? *g.geo_from : g;
            launch_synth_ls(p.get("Hc"), p.ldD, ls_end, g0p.bufs.at("bsc").p, synth_nord_pad(nOrd), p.get<double>("hrir_azi"), p.get<double>("hrir_zen"),
                            p.get<double>("mic_azi"), p.get<double>("mic_zen"), p.get<int>("smap"), (int)g.D, (int)g.d.nmics, g.P, gf, ls_end, p.get("Usw"), st, true);
            launch_synth_rows(p.get("Usw"), synth_ls_chunks((int)g.D), g0p.bufs.at("Pm").p, g0p.bufs.at("Mw").p, g.C, (int)g.d.nmics, gf, ls_end, g.P, p.get("W"), st, true);
        } else
        launch_ls_gram(p.get("Hc"), p.ldD, ls_end, g.get<cplx>("G") - (int64_t)g.g0 * g_stride, g_stride, g.ldD, g.get("Mw"), (int)g.D, g.C, g.P, gf,
                       ls_end, p.get("W"), st);
    }
}
// One stream for the whole batch (the subjects' stages are short and the sweep chain is what bounds a batch of HRIR sets), so
// that the stages before and after the sweep are two single-stream graphs: issued eagerly, the ~250 launches of a 16-set batch
// cost 11 ms of host time (measured: 1380 sets/s whatever the number of batches in flight).
void batch_geo_stage(emagls_batch& b, int part) {
    emagls_plan& p0 = *b.plans[0];
    std::vector<hipStream_t> keep;
    for (auto* p : b.plans) { keep.push_back(p->stream); p->stream = b.stream; }
    auto restore = [&] { for (size_t j = 0; j < b.plans.size(); ++j) b.plans[j]->stream = keep[j]; };
    const int nsub = (int)b.plans.size() - 1;
    try {
        if (part == 0) {
            plan_pre_stage(p0);
            if (b.lanes && nsub > 0) {
                // lane batch: the subjects' stages are ONE launch per kernel for all of them (plans 1.. at the arena stride).  The
                // few geometry operands those kernels read (conj(Y), R, the Householder-route factors, M_k and G_k of the
                // least-squares bins: ~40 MB) are copied into the subjects' own slots first, so that every pointer of a launch
                // moves by the same stride; the large ones (G_k, M_k of the swept bins) are only read by the sweep, through
                // plan 0's pointers.
                emagls_plan& g = p0;
                emagls_plan& p1 = *b.plans[1];
                const bool cb = g.cplx_basis;
                const int gf = g.gram_from, hh_end = g.hh_end, ldSh = g.ldS_h;
                const int ls_end = std::min(g.kcut0, g.P);
                const size_t g_stride_b = sizeof(cplx) * (size_t)g.C * g.ldD;
                auto bc = [&](const char* name, size_t off, size_t bytes) {
                    if (!g.has(name) || bytes == 0) return;
                    bytes = std::min(bytes, g.bufs[name].bytes - off);
                    launch_broadcast_lanes(g.get<char>(name) + off, bytes, b.stride, nsub, b.stream);
                };
                if (hh_end > 1) {
                    bc("Yc", 0, g.bufs["Yc"].bytes);
                    bc(cb ? "R" : "Rc", 0, g.bufs[cb ? "R" : "Rc"].bytes);
                    bc("Vws", 0, sizeof(cplx) * (size_t)(hh_end - 1) * g.C * ldSh);
                    bc("Nw", 0, sizeof(cplx) * (size_t)(hh_end - 1) * g.C * g.C);
                    bc("tauw", 0, sizeof(double) * (size_t)(hh_end - 1) * g.C);
                    bc("cond_ok", 0, sizeof(double) * (size_t)g.P);
                }
                if (gf > 0 && gf < ls_end) {
                    bc("G", (size_t)(gf - g.g0) * g_stride_b, (size_t)(ls_end - gf) * g_stride_b);
                    bc("Mw", 0, sizeof(cplx) * (size_t)ls_end * g.C * g.C);
                }
                BatchScope sc(nsub, b.stride);
                emagls_subject_prologue(p1, p0);
                p1.geo_from = &p0;   // (synthesising designs: plan 0's coefficients, Pm and M_k for every lane)
                try { emagls_subject_rows(p1, p1); } catch (...) { p1.geo_from = nullptr; throw; }
                p1.geo_from = nullptr;
                if (p0.synth)   // every set's own start value of the microphone-domain chain, on plan 0's Pm
                    launch_synth_winit(p1.get("W"), p0.get("Pm"), p0.C, (int)p0.d.nmics, std::max(p0.kcut0, 1), p0.P, p1.get("Winit"), b.stream, true);
            } else {
                for (size_t j = 1; j < b.plans.size(); ++j) {
                    emagls_subject_prologue(*b.plans[j], p0);
                    emagls_subject_rows(*b.plans[j], p0);
                    if (p0.synth)
                        launch_synth_winit(b.plans[j]->get("W"), p0.get("Pm"), p0.C, (int)p0.d.nmics, std::max(p0.kcut0, 1), p0.P, b.plans[j]->get("Winit"),
                                           b.stream, true);
                }
            }
        } else if (b.lanes) {
            BatchScope sc((int)b.plans.size(), b.stride);
            p0.geo_from = &p0;   // (the filters' rows of every lane from plan 0's Pm and M_k)
            try { emagls_post_sweep(p0); } catch (...) { p0.geo_from = nullptr; throw; }
            p0.geo_from = nullptr;
        } else {
            for (auto* p : b.plans) {
                p->geo_from = &p0;
                try { emagls_post_sweep(*p); } catch (...) { p->geo_from = nullptr; throw; }
                p->geo_from = nullptr;
            }
        }
    } catch (...) {
        restore();
        throw;
    }
    restore();
}
void batch_execute_geo(emagls_batch& b) {
    emagls_plan& p0 = *b.plans[0];
    const bool replay = b.use_graph && b.eager_runs >= 1;
    if (replay && !b.graph_exec) {
        capture_into(b.stream, &b.graph, &b.graph_exec, [&] { batch_geo_stage(b, 0); });
        capture_into(b.stream, &b.post_graph, &b.post_exec, [&] { batch_geo_stage(b, 2); });
    }
    b.used = 0;
    if (replay) HIP_CHECK(hipGraphLaunch(b.graph_exec, b.stream)); else batch_geo_stage(b, 0);
    batch_sweep_stage(b);   // (never captured: see SweepChain)
    if (replay) HIP_CHECK(hipGraphLaunch(b.post_exec, b.stream)); else batch_geo_stage(b, 2);
    for (auto* p : b.plans) {
        p->executed = true;
        p->sweep_launches = p0.sweep_persist ? 1 : p0.P - std::max(p0.kcut0, 1);
    }
    if (!replay) ++b.eager_runs;
}

// ---------------------------------------------------------------------------------------------
// MagLS / MagLS-2D batches (lib/getMagLsFilters.m:30, getMagLsFilters2D.m:1 in a loop over HRIR sets): every plan's stages before
// and after the sweep on the batch's stream (two single-stream graphs) and ONE resident sweep launch for all designs instead
// of one per design.  With geometry sharing (same grid: SH matrix, its Cholesky factor, pinv(Y_conj), the sweep's operands
// G = Y_conj and M = R^-1 R^-H are the same for every set) plan 0 computes that side and the other plans run their HRIR
// prologue and least-squares bins on it.
// ---------------------------------------------------------------------------------------------
void batch_magls_decide_sharing(emagls_batch& b) {
    bool share = false;
    emagls_plan& p0 = *b.plans[0];
    if (b.geo_want && b.plans.size() > 1 && !p0.custom_basis && !p0.diffuse && p0.sweep_persist) {
        uint64_t ver = 0;
        for (auto* p : b.plans) ver = ver * 1000003ull + p->atf_side_version;
        if (ver != b.geo_checked_version) {
            if (!b.cmp_flag) HIP_CHECK(hipMalloc(&b.cmp_flag, 16));
            HIP_CHECK(hipStreamSynchronize(b.stream));
            HIP_CHECK(hipMemsetAsync(b.cmp_flag, 0, 16, b.stream));
            for (size_t j = 1; j < b.plans.size(); ++j)
                for (const char* name : {"hrir_azi", "hrir_zen"})
                    launch_compare_words(p0.get(name), b.plans[j]->get(name), p0.bufs[name].bytes, b.cmp_flag, b.stream);
            int differ = 0;
            HIP_CHECK(hipMemcpyAsync(&differ, b.cmp_flag, sizeof differ, hipMemcpyDeviceToHost, b.stream));
            HIP_CHECK(hipStreamSynchronize(b.stream));
            b.geo_checked_version = ver;
            b.geo_inputs_same = differ == 0;
        }
        share = b.geo_inputs_same;
        for (auto* p : b.plans) share = share && !p->custom_basis && p->d.fs == p0.d.fs;
    }
    if (share != b.geo_share) {
        for (auto* p : b.plans) drop_plan_graphs(*p);
        drop_batch_graphs(b);
        b.geo_share = share;
    }
}
void batch_magls_stage(emagls_batch& b, int part) {
    emagls_plan& g = *b.plans[0];
    std::vector<hipStream_t> keep;
    for (auto* p : b.plans) { keep.push_back(p->stream); p->stream = b.stream; }
    auto restore = [&] { for (size_t j = 0; j < b.plans.size(); ++j) b.plans[j]->stream = keep[j]; };
    try {
        if (part == 0) {
            for (size_t j = 0; j < b.plans.size(); ++j) {
                emagls_plan& p = *b.plans[j];
                if (j == 0 || !b.geo_share) { plan_pre_stage(p); continue; }
                // a subject of plan 0's grid: spectra, least-squares bins on plan 0's pinv(Y_conj)
                p.stage_names.clear();
                launch_zero(p.get("flag"), sizeof(int) * NFLAG, b.stream);
                launch_zero(p.get("W"), p.bufs["W"].bytes, b.stream);
                stage_prologue(p, 0, nullptr, p.D);
                launch_ls_apply(p.get("Hc"), p.ldD, std::min(p.kcut0, p.P), g.get("Ypinv"), g.cplx_basis, g.ldD, (int)p.D, p.C, p.P, 0,
                                std::min(p.kcut0, p.P), p.get("W"), b.stream);
            }
        } else {
            for (auto* p : b.plans) magls_post_sweep(*p);
        }
    } catch (...) {
        restore();
        throw;
    }
    restore();
}
void batch_execute_magls(emagls_batch& b) {
    emagls_plan& p0 = *b.plans[0];
    for (auto* p : b.plans)
        if (!p->have_hrirs || (p->custom_basis ? !p->have_basis : !p->have_hrir_grid))
            throw Error(EMAGLS_ERR_ARG, "every plan of the batch needs its grid (or SH matrix) and HRIRs");
    if (p0.d.kind == EMAGLS_KIND_LS) {
        // getLsFilters (lib/getLsFilters.m:30-34) has no sweep: wLs = h pinv(Y).  Sets on one grid: pinv(Y) once (plan 0), one
        // small product per set; otherwise every plan's own pipeline, all on the batch's stream.
        bool keep_persist = p0.sweep_persist;
        p0.sweep_persist = true;                 // (the sharing decision only asks for it on behalf of the sweep; LS has none)
        try { batch_magls_decide_sharing(b); } catch (...) { p0.sweep_persist = keep_persist; throw; }
        p0.sweep_persist = keep_persist;
        for (size_t j = 0; j < b.plans.size(); ++j) {
            emagls_plan& p = *b.plans[j];
            hipStream_t keep = p.stream;
            p.stream = b.stream;
            try {
                p.stage_names.clear();
                launch_zero(p.get("flag"), sizeof(int) * NFLAG, b.stream);
                if (j == 0 || !b.geo_share) execute_ls(p);
                else launch_ls_filters(p.get<double>("hL"), p.get<double>("hR"), p.d.nsamp, (int)p.D, p0.get("Ypinv"), p0.cplx_basis, p0.ldD, p.C,
                                       p.get("wL"), p.get("wR"), b.stream);
            } catch (...) { p.stream = keep; throw; }
            p.stream = keep;
            p.executed = true;
        }
        return;
    }
    bool persist = true;
    for (auto* p : b.plans) persist = persist && p->sweep_persist;
    if (!persist) {   // (an ill-conditioned basis or a sweep that did not become resident: the designs one at a time, launch-per-bin sweeps)
        if (b.geo_share) { for (auto* p : b.plans) drop_plan_graphs(*p); drop_batch_graphs(b); b.geo_share = false; }
        for (auto* p : b.plans) {
            hipStream_t keep = p->stream;
            p->stream = b.stream;
            try { plan_execute(*p); } catch (...) { p->stream = keep; throw; }
            p->stream = keep;
        }
        return;
    }
    batch_magls_decide_sharing(b);
    const bool replay = b.use_graph && b.eager_runs >= 1;
    if (replay && !b.graph_exec) {
        capture_into(b.stream, &b.graph, &b.graph_exec, [&] { batch_magls_stage(b, 0); });
        capture_into(b.stream, &b.post_graph, &b.post_exec, [&] { batch_magls_stage(b, 2); });
    }
    b.used = 0;
    if (replay) HIP_CHECK(hipGraphLaunch(b.graph_exec, b.stream)); else batch_magls_stage(b, 0);
    batch_sweep_stage(b);   // (never captured: see SweepChain)
    if (replay) HIP_CHECK(hipGraphLaunch(b.post_exec, b.stream)); else batch_magls_stage(b, 2);
    for (auto* p : b.plans) { p->executed = true; p->sweep_launches = 1; }
    (void)p0;
    if (!replay) ++b.eager_runs;
}

void batch_execute(emagls_batch& b) {
    for (auto* p : b.plans)
        if (!p) throw Error(EMAGLS_ERR_ARG, "a plan of this batch has been destroyed");
    if (b.atf) { batch_execute_atf(b); return; }
    if (b.magls) { batch_execute_magls(b); return; }
    for (auto* p : b.plans)
        if (!p->have_hrirs || (p->custom_basis ? !p->have_basis : (!p->have_hrir_grid || !p->have_mic_grid)))
            throw Error(EMAGLS_ERR_ARG, "every plan of the batch needs its grids (or SH matrices) and HRIRs");
    batch_geo_decide_sharing(b);
    if (b.geo_share) { batch_execute_geo(b); return; }
    if (b.lanes) {
        batch_execute_lanes(b);
        return;
    }
    const bool replay = b.use_graph && b.eager_runs >= 1;
    if (replay && !b.plans[0]->pre_exec) {
        for (auto* p : b.plans) capture_into(p->stream, &p->pre_graph, &p->pre_exec, [&] { plan_pre_stage(*p); });
        if (!b.plans[0]->sweep_persist) capture_into(b.stream, &b.graph, &b.graph_exec, [&] { batch_sweep_stage(b); });
    }
    b.used = 0;
    // the previous sweep of this batch must be done before a plan's buffers are rewritten
    for (auto* p : b.plans) b.depend(p->stream, b.stream);
    for (auto* p : b.plans) {
        if (replay) HIP_CHECK(hipGraphLaunch(p->pre_exec, p->stream)); else plan_pre_stage(*p);
        b.depend(b.stream, p->stream);
    }
    if (b.plans[0]->sweep_persist) batch_sweep_stage(b);   // (never captured: see SweepChain)
    else if (replay) HIP_CHECK(hipGraphLaunch(b.graph_exec, b.stream)); else batch_sweep_stage(b);
    emagls_plan& p0 = *b.plans[0];
    for (auto* p : b.plans) {
        b.depend(p->stream, b.stream);
        emagls_post_sweep(*p);
        b.depend(b.stream, p->stream);  // batch stream completion == all results ready
        p->executed = true;
        p->sweep_launches = p0.P - std::max(p0.kcut0, 1);
    }
    if (!replay) ++b.eager_runs;
}

void plan_execute(emagls_plan& p);
void batch_execute(emagls_batch& b);

void drop_plan_graphs(emagls_plan& p) {
    if (p.graph_exec) { HIP_CHECK(hipGraphExecDestroy(p.graph_exec)); p.graph_exec = nullptr; }
    if (p.graph) { HIP_CHECK(hipGraphDestroy(p.graph)); p.graph = nullptr; }
    if (p.pre_exec) { HIP_CHECK(hipGraphExecDestroy(p.pre_exec)); p.pre_exec = nullptr; }
    if (p.pre_graph) { HIP_CHECK(hipGraphDestroy(p.pre_graph)); p.pre_graph = nullptr; }
    p.eager_runs = 0;
}
void drop_batch_graphs(emagls_batch& b) {
    if (b.graph_exec) { HIP_CHECK(hipGraphExecDestroy(b.graph_exec)); b.graph_exec = nullptr; }
    if (b.graph) { HIP_CHECK(hipGraphDestroy(b.graph)); b.graph = nullptr; }
    if (b.post_exec) { HIP_CHECK(hipGraphExecDestroy(b.post_exec)); b.post_exec = nullptr; }
    if (b.post_graph) { HIP_CHECK(hipGraphDestroy(b.post_graph)); b.post_graph = nullptr; }
    if (b.graph2_exec) { HIP_CHECK(hipGraphExecDestroy(b.graph2_exec)); b.graph2_exec = nullptr; }
    if (b.graph2) { HIP_CHECK(hipGraphDestroy(b.graph2)); b.graph2 = nullptr; }
    for (int i = 0; i < 2; ++i) {
        if (b.graphx_exec[i]) { HIP_CHECK(hipGraphExecDestroy(b.graphx_exec[i])); b.graphx_exec[i] = nullptr; }
        if (b.graphx[i]) { HIP_CHECK(hipGraphDestroy(b.graphx[i])); b.graphx[i] = nullptr; }
    }
    for (int i = 0; i < 4; ++i) {
        if (b.graph_hh_exec[i]) { HIP_CHECK(hipGraphExecDestroy(b.graph_hh_exec[i])); b.graph_hh_exec[i] = nullptr; }
        if (b.graph_hh[i]) { HIP_CHECK(hipGraphDestroy(b.graph_hh[i])); b.graph_hh[i] = nullptr; }
    }
    b.eager_runs = 0;
}
// Device-side status words of a design: [0] Cholesky pivot, [1] persistent sweep gave up waiting, [2] a Gram-route bin was
// worse conditioned than the kr estimate promised ([3] = the highest such bin), [4] MagLS: the SH basis is too ill-conditioned
// for the inverse form M = R^-1 R^-H of the persistent sweep (the reference's pinv would drop singular values), [5] LS / MagLS
// above 32 channels: basis too ill-conditioned for the Gram-inverse form of pinv (fatal: no SVD route at that width).
// [1], [2] and [4] are recoverable: the design is re-run without the feature.  [1] and [2] stick to the plan (a residency or
// conditioning property of the shape); [4] is a property of THIS call's grid, so the launch-per-bin sweep only serves the
// re-run and a cached plan tries the persistent form again on its next call.
// Returns true when the design has to be executed again; throws when a flag cannot be recovered from.
bool plan_recover(emagls_plan& p, const int* flag, bool apply) {
    bool redo = false;
    if (flag[2]) {
        // flag[3] = the highest Gram-route bin whose condition number exceeded the limit: the route restarts behind it (the
        // Householder route then covers more bins and, at their higher kr, more orders: plan_routes refuses beyond its tile)
        if (p.gram_from == 0 || flag[3] < p.gram_from)
            throw Error(EMAGLS_ERR_NUMERIC, "internal: Gram-route conditioning flag outside the route (stale graph)");
        if (p.d.kind == EMAGLS_KIND_FROM_ATF && p.C > 8)
            throw Error(EMAGLS_ERR_UNSUPPORTED, "the ATF matrices of some bins are too ill-conditioned for the Gram route (cond > 3e4) and the "
                                                "dense route holds at most 8 microphones in this build");
        if (p.d.kind == EMAGLS_KIND_FROM_ATF && p.Dm > 4096)
            throw Error(EMAGLS_ERR_UNSUPPORTED, "the ATF matrices of some bins are too ill-conditioned for the Gram route (cond > 3e4) and the "
                                                "dense route holds at most 4096 matched directions in this build");
        if (apply && p.d.kind == EMAGLS_KIND_FROM_ATF) {
            // measured ATFs: the bins up to the offending one take the dense route (QR + Jacobi of the matched ATF matrix itself)
            p.gram_from = flag[3] + 1 < p.P ? flag[3] + 1 : 0;
        } else if (apply) {
            p.gram_floor = std::max(p.gram_floor, flag[3] + 1);
            plan_routes(p);
            plan_alloc_routes(p);
            HIP_CHECK(hipStreamSynchronize(p.stream));
        }
        redo = true;
    }
    if (flag[4]) {
        if (!p.sweep_persist) throw Error(EMAGLS_ERR_NUMERIC, "internal: MagLS conditioning flag without the persistent sweep");
        if (apply) { p.sweep_persist = false; p.persist_suspended = true; if (p.synth_want) { plan_alloc_routes(p); HIP_CHECK(hipStreamSynchronize(p.stream)); } }
        redo = true;
    }
    if (flag[1]) {
        // not every workgroup of the persistent sweep became resident (CUs held by another process, partitioned device):
        // the launch-per-bin sweep needs no co-residency
        if (!p.sweep_persist) throw Error(EMAGLS_ERR_HIP, "phase sweep: a workgroup timed out waiting for its peers' partial sums");
        if (apply) { p.sweep_persist = false; if (p.synth_want) { plan_alloc_routes(p); HIP_CHECK(hipStreamSynchronize(p.stream)); } }
        redo = true;
    }
    return redo;
}
void throw_fatal_flags(const int* flag) {
    if (flag[1]) throw Error(EMAGLS_ERR_HIP, "phase sweep: a workgroup timed out waiting for its peers' partial sums");
    if (flag[2]) throw Error(EMAGLS_ERR_NUMERIC, "per-bin factorisation: ill-conditioned bin on the Gram route after the re-run");
    if (flag[5])
        throw Error(EMAGLS_ERR_UNSUPPORTED, "the SH basis of this order is too ill-conditioned on the HRIR grid for the 33..64-channel path "
                                            "(cond > 1e4: pinv would need the SVD route, which stops at 32 channels in this build)");
    if (flag[0])
        throw Error(EMAGLS_ERR_NUMERIC,
                    "SH Gram matrix of the HRIR grid is not positive definite (the grid cannot resolve the required SH order)");
}
// re-run a whole batch after one of its designs raised a recoverable flag: in lane mode all designs share the captured
// graphs, so every plan of the batch changes its configuration together
// Lane mode needs plans of identical shape (same buffers of the same sizes, same derived constants).  Their
// buffers are moved into one arena at a constant stride; the plans keep working on their own afterwards.
// Lane mode launches every kernel once for all designs with the routes of the first one, so the designs of a batch get
// common routes first: the latest start of the Gram route and the most Householder-route orders any of them asks for (both are
// valid for every member: the Householder route is accurate anywhere, more orders only add terms below the noise floor).
// Designs of one simulation-order class but different radii (BASELINE config 4) differ by a bin or an order here.
bool batch_unify_routes_once(emagls_batch& b);
void batch_unify_routes(emagls_batch& b) {
    // (moving a design's route boundary changes the orders its Householder bins need: repeat until nothing moves)
    for (int it = 0; it < 4 && batch_unify_routes_once(b); ++it) {}
}
// returns true when a plan's routes were changed
bool batch_unify_routes_once(emagls_batch& b) {
    int gf = 0, nh = 0;
    bool differ = false;
    for (auto* p : b.plans) {
        if (p->gram_from <= 0 || p->d.kind == EMAGLS_KIND_EMA_SH) return false;
        differ = differ || p->gram_from != b.plans[0]->gram_from || p->n_h != b.plans[0]->n_h;
        gf = std::max(gf, p->gram_from);
        nh = std::max(nh, p->n_h);
    }
    if (!differ) return false;
    for (auto* p : b.plans) {
        if (p->gram_from == gf && p->n_h == nh) continue;
        const int keep_floor = p->gram_floor, keep_nh = p->nh_floor;
        try {
            p->gram_floor = std::max(p->gram_floor, gf);
            p->nh_floor = nh;
            plan_routes(*p);
            plan_alloc_routes(*p);
        } catch (const Error& e) {   // (e.g. more Householder-route orders than the register tile holds: keep the plan's own routes)
            if (getenv("EMAGLS_DEBUG_LANES")) fprintf(stderr, "common routes refused: %s\n", e.what());
            p->gram_floor = keep_floor; p->nh_floor = keep_nh;
            plan_routes(*p);
            plan_alloc_routes(*p);
            return false;
        }
        if (p->graph_exec) { HIP_CHECK(hipGraphExecDestroy(p->graph_exec)); p->graph_exec = nullptr; }
        if (p->graph) { HIP_CHECK(hipGraphDestroy(p->graph)); p->graph = nullptr; }
        if (p->pre_exec) { HIP_CHECK(hipGraphExecDestroy(p->pre_exec)); p->pre_exec = nullptr; }
        if (p->pre_graph) { HIP_CHECK(hipGraphDestroy(p->pre_graph)); p->pre_graph = nullptr; }
        p->eager_runs = 0;
    }
    return true;
}

// One sweep launch serves every design of a batch: the synthesising form only when all of them qualify
void batch_unify_synth(emagls_batch& b) {
    bool all = true, any = false;
    for (auto* p : b.plans) { all = all && p->synth; any = any || p->synth; }
    if (all || !any) return;
    for (auto* p : b.plans)
        if (p->synth) { p->synth_block = true; plan_alloc_routes(*p); HIP_CHECK(hipStreamSynchronize(p->stream)); drop_plan_graphs(*p); }
}
// Can the resident sweep of the form the batch will launch keep all its workgroups on the device?  Decided before any launch, from the
// runtime's occupancy of that kernel variant (a sweep that cannot be resident would wait for its peers until the time-out); re-evaluated
// whenever the form changes (batch_redo).  A batch that does not fit takes one launch per bin.
void batch_decide_residency(emagls_batch& b) {
    emagls_plan& f0 = *b.plans[0];
    const int n = (int)b.plans.size();
    if (f0.d.kind == EMAGLS_KIND_LS) return;
    bool all_persist = true;
    for (auto* p : b.plans) all_persist = all_persist && p->sweep_persist;
    if (!all_persist) return;
    const int64_t Dh0 = f0.d.kind == EMAGLS_KIND_FROM_ATF ? f0.Dm : f0.D;
    const bool fits = f0.synth ? (reg_sweep_wanted(b.plans.data(), n) || (n <= SWEEP_MULTI_MAX && synth_sweep_fits((int)Dh0, (int)f0.d.nmics, f0.simOrder + 1, n)))
                               : (n <= SWEEP_MULTI_MAX && persist_sweep_fits((int)Dh0, f0.C, n));
    if (fits) return;
    for (auto* p : b.plans) {
        p->sweep_persist = false;
        if (p->synth_want) { plan_alloc_routes(*p); HIP_CHECK(hipStreamSynchronize(p->stream)); }
    }
}

void batch_try_lanes(emagls_batch& b) {
    if (const char* e = getenv("EMAGLS_BATCH_LANES")) if (e[0] == '0') return;
    emagls_plan& q = *b.plans[0];
    if (!q.sweep_persist) return;
    for (auto* p : b.plans)
        if (p->S != q.S || p->simOrder != q.simOrder || p->d.kind != q.d.kind || p->C != q.C || p->P != q.P) return;
    trace_mark("lanes: start");
    batch_unify_routes(b);
    trace_mark("lanes: routes unified");
    batch_unify_synth(b);
    const bool dbg = getenv("EMAGLS_DEBUG_LANES") != nullptr;
    for (auto* p : b.plans) {
        if (p->S != q.S || p->simOrder != q.simOrder || p->nOut != q.nOut || p->nfft != q.nfft || p->ldS != q.ldS || p->ldD != q.ldD ||
            p->Dpad != q.Dpad || p->k_cut != q.k_cut || p->cplx_basis != q.cplx_basis || p->out_cplx != q.out_cplx ||
            p->d.kind != q.d.kind || p->d.nsamp != q.d.nsamp || p->d.nmics != q.d.nmics || p->d.len != q.d.len || p->d.order != q.d.order ||
            p->bufs.size() != q.bufs.size()) {
            if (dbg) fprintf(stderr, "lanes refused: shape fields differ (bufs %zu vs %zu, gram_from %d vs %d, n_h %d vs %d)\n", p->bufs.size(),
                             q.bufs.size(), p->gram_from, q.gram_from, p->n_h, q.n_h);
            return;
        }
        auto it = q.bufs.begin();
        for (auto& kv : p->bufs) {
            if (kv.first != it->first || kv.second.bytes != it->second.bytes) {
                if (dbg) fprintf(stderr, "lanes refused: buffer %s %zu vs %s %zu (gram_from %d vs %d, hh_end %d vs %d, n_h %d vs %d)\n", kv.first.c_str(),
                                 kv.second.bytes, it->first.c_str(), it->second.bytes, p->gram_from, q.gram_from, p->hh_end, q.hh_end, p->n_h, q.n_h);
                return;
            }
            ++it;
        }
    }
    size_t stride = 0;
    std::vector<size_t> off;
    for (auto& kv : q.bufs) {
        off.push_back(stride);
        stride += (kv.second.bytes + 255) / 256 * 256;
    }
    stride = (stride + 4095) / 4096 * 4096;
    trace_mark("lanes: shapes compared");
    auto arena = std::make_shared<Arena>();
    {
        const size_t need = stride * b.plans.size();
        arena->base = BlockPool::get().take(need, &arena->bytes, BlockPool::size_class(need + need / 8));
    }
    for (size_t j = 0; j < b.plans.size(); ++j) {   // (one launch per 96 buffers: move_buffers_kernel)
        emagls_plan& p = *b.plans[j];
        size_t i = 0;
        BufferMoves mv{};
        for (auto& kv : p.bufs) {
            char* dst = static_cast<char*>(arena->base) + j * stride + off[i++];
            if ((reinterpret_cast<uintptr_t>(kv.second.p) & 15) != 0) { HIP_CHECK(hipMemcpyAsync(dst, kv.second.p, kv.second.bytes, hipMemcpyDeviceToDevice, b.stream)); continue; }
            mv.src[mv.n] = kv.second.p; mv.dst[mv.n] = dst; mv.bytes[mv.n] = kv.second.bytes;
            if (++mv.n == 96) { launch_move_buffers(mv, b.stream); mv.n = 0; }
        }
        launch_move_buffers(mv, b.stream);
    }
    trace_mark("lanes: arena taken, copies enqueued");
    HIP_CHECK(hipStreamSynchronize(b.stream));   // (every plan's streams were synchronised by the caller: the buffers are final)
    trace_mark("lanes: copies done");
    for (size_t j = 0; j < b.plans.size(); ++j) {
        emagls_plan& p = *b.plans[j];
        size_t i = 0;
        for (auto& kv : p.bufs) {
            if (kv.second.owned) HIP_CHECK(hipFree(kv.second.p));
            kv.second.p = static_cast<char*>(arena->base) + j * stride + off[i++];
            kv.second.owned = false;
        }
        p.release_slabs();
        p.arena = arena;  // (a previous arena is released when its last plan has moved out)
        // the captured graphs hold the old addresses
        if (p.graph_exec) { HIP_CHECK(hipGraphExecDestroy(p.graph_exec)); p.graph_exec = nullptr; }
        if (p.graph) { HIP_CHECK(hipGraphDestroy(p.graph)); p.graph = nullptr; }
        if (p.pre_exec) { HIP_CHECK(hipGraphExecDestroy(p.pre_exec)); p.pre_exec = nullptr; }
        if (p.pre_graph) { HIP_CHECK(hipGraphDestroy(p.pre_graph)); p.pre_graph = nullptr; }
        p.eager_runs = 0;
    }
    HIP_CHECK(hipDeviceSynchronize());
    b.lanes = true;
    b.stride = stride;
    {   // more than 8 designs: two lane groups before the sweep (EMAGLS_BATCH_GROUPS=1 keeps one launch sequence for all lanes, 3 / 4
        // allow groups of 8 for 17 ... 32 designs: measured with 32-design batches, 3125 / 3345 sets/s at 128 / 512 steps with four
        // groups against 3296 / 3499 with two, and the same 2070 at 20 steps)
        const char* e = getenv("EMAGLS_BATCH_GROUPS");
        const int cap = e ? std::max(1, std::min(4, atoi(e))) : 2;
        b.groups = std::max(1, std::min(cap, (int)ceil_div((int64_t)b.plans.size(), 8)));
    }
}

void batch_redo(emagls_batch& b, const std::vector<int>& flags) {
    int any[NFLAG] = {};
    for (size_t j = 0; j < b.plans.size(); ++j) for (int i = 0; i < NFLAG; ++i) any[i] = std::max(any[i], flags[NFLAG * j + i]);
    bool moved = false;
    for (auto* q : b.plans) {
        const int64_t before = q->total_bytes;
        plan_recover(*q, any, true);
        moved = moved || q->total_bytes != before;
        drop_plan_graphs(*q);
    }
    drop_batch_graphs(b);
    {
        const std::vector<int64_t> before = [&] { std::vector<int64_t> v; for (auto* q : b.plans) v.push_back(q->total_bytes); return v; }();
        batch_unify_synth(b);
        batch_decide_residency(b);   // (the form may have changed: the residency of the kernel that will be launched)
        for (size_t j = 0; j < b.plans.size(); ++j) moved = moved || b.plans[j]->total_bytes != before[j];
    }
    if (b.lanes && moved) {   // re-allocated buffers left the arena: lane mode needs them at the common stride again
        b.lanes = false;
        batch_try_lanes(b);
    }
    batch_execute(b);
    HIP_CHECK(hipStreamSynchronize(b.stream));
}
std::vector<int> batch_read_flags(emagls_batch& b) {
    const size_t n = b.plans.size();
    std::vector<int> flags(NFLAG * n, 0);
    for (size_t j = 0; j < n; ++j)
        HIP_CHECK(hipMemcpyAsync(&flags[NFLAG * j], b.plans[j]->get("flag"), NFLAG * sizeof(int), hipMemcpyDeviceToHost, b.stream));
    HIP_CHECK(hipStreamSynchronize(b.stream));
    return flags;
}
void plan_check_flags(emagls_plan& p) {
    int flag[NFLAG] = {};
    HIP_CHECK(hipMemcpy(flag, p.get("flag"), sizeof flag, hipMemcpyDeviceToHost));
    if (plan_recover(p, flag, false)) {
        if (p.owner) {   // a member of a batch: the batch re-runs as a whole (its graphs cover every member)
            emagls_batch& b = *p.owner;
            for (auto* q : b.plans) if (!q) throw Error(EMAGLS_ERR_ARG, "a plan of this batch has been destroyed");
            batch_redo(b, batch_read_flags(b));
        } else {
            plan_recover(p, flag, true);
            drop_plan_graphs(p);
            plan_execute(p);
            HIP_CHECK(hipStreamSynchronize(p.stream));
        }
        HIP_CHECK(hipMemcpy(flag, p.get("flag"), sizeof flag, hipMemcpyDeviceToHost));
        if (plan_recover(p, flag, false)) {   // e.g. first the Gram route, then the persistent sweep
            if (p.owner) batch_redo(*p.owner, batch_read_flags(*p.owner));
            else { plan_recover(p, flag, true); drop_plan_graphs(p); plan_execute(p); HIP_CHECK(hipStreamSynchronize(p.stream)); }
            HIP_CHECK(hipMemcpy(flag, p.get("flag"), sizeof flag, hipMemcpyDeviceToHost));
        }
    }
    if (p.persist_suspended) {   // (the re-run on the launch-per-bin sweep is done: the next call starts on the persistent form again)
        p.persist_suspended = false;
        p.sweep_persist = true;
    }
    throw_fatal_flags(flag);
    p.geo_done_version = p.geo_run_version;   // (clean: a later set on the same grids may keep this run's geometry stages)
}

}  // namespace
hipStream_t emagls::pool_stream_take() { return StreamPool::get().take(); }
void emagls::pool_stream_give(hipStream_t st) { StreamPool::get().give(st); }
int emagls::guarded_call(const std::function<void()>& f) {
    try {
        f();
        return EMAGLS_OK;
    } catch (const Error& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return EMAGLS_ERR_HIP;
    }
}
namespace {
template <typename F> int guarded(F&& f) {
    try {
        f();
        return EMAGLS_OK;
    } catch (const Error& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return EMAGLS_ERR_HIP;
    }
}

// ---------------------------------------------------------------------------------------------
// One-shot entry points (what the MEX shim binds: lib/get*Filters.m signatures, host arrays in, filters out).
// A shape-keyed cache keeps the plans of recent calls alive -- device buffers, captured graphs, the routes a
// conditioning check may have moved -- so that a repeated design costs the uploads, one replay and the download instead
// of ~1.5 GB of hipMalloc and an eager first execute.  emagls_cache_clear() (mexAtExit) releases everything.
// ---------------------------------------------------------------------------------------------
struct CachedPlan {
    std::unique_ptr<emagls_plan> plan;
    emagls_design_desc desc{};
    int device = 0;
    bool busy = false;
    uint64_t last_use = 0;
};
std::mutex g_cache_mu;
std::vector<CachedPlan> g_cache;
uint64_t g_cache_tick = 0;

size_t plan_cache_capacity() {
    static const size_t cap = [] { const char* e = getenv("EMAGLS_PLAN_CACHE"); return e ? (size_t)std::max(0, atoi(e)) : (size_t)4; }();
    return cap;
}
bool same_desc(const emagls_design_desc& a, const emagls_design_desc& b) {
    return a.kind == b.kind && a.basis == b.basis && a.order == b.order && a.fs == b.fs && a.len == b.len && a.nsamp == b.nsamp &&
           a.ndirs == b.ndirs && a.mic_radius == b.mic_radius && a.nmics == b.nmics && a.f_trans == b.f_trans &&
           a.atf_taps == b.atf_taps && a.natf == b.natf && a.custom_basis == b.custom_basis && a.diffuseness == b.diffuseness &&
           a.sim_order_pad == b.sim_order_pad;
}

int one_shot(const emagls_design_desc& desc, const double* hL, const double* hR, const double* azi, const double* zen,
             const double* mic_azi, const double* mic_zen, const double* atf, const double* atf_azi, const double* atf_zen,
             void* wL, void* wR, double* mean_dev, const void* Y_hrir = nullptr, const void* Y_mic = nullptr) {
    return guarded([&] {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        emagls_plan* p = nullptr;
        std::unique_ptr<emagls_plan> fresh;
        bool cached = false;
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (auto& c : g_cache)
                if (!c.busy && c.device == dev && same_desc(c.desc, desc)) { c.busy = true; p = c.plan.get(); cached = true; break; }
        }
        if (!p) {
            fresh.reset(new emagls_plan);
            fresh->d = desc;
            plan_setup(*fresh);
            if (array_kind(desc.kind)) {
                // One stream: the independent branches of a single design could fork onto side streams (3 streams: 3.55 instead of
                // ~4 ms in a plan), but a MULTI-stream capture is what both hipGraphLaunch crashes of this round had in common
                // (StreamPool above; again with the pool in place after a lane batch of the same shape) and a cached one-shot plan
                // is captured and replayed inside whatever session the caller runs.  EMAGLS_ONESHOT_STREAMS=3 restores the forks.
                const char* e = getenv("EMAGLS_ONESHOT_STREAMS");
                fresh->nstreams = e ? std::max(1, std::min(3, atoi(e))) : 1;
                fresh->alone = true;   // (a one-shot call has the device to itself: the orthonormal route of the low bins runs next to the sweep)
            }
            p = fresh.get();
        }
        auto release = [&](bool ok) {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            if (cached) {
                for (size_t i = 0; i < g_cache.size(); ++i)
                    if (g_cache[i].plan.get() == p) {
                        if (ok) { g_cache[i].busy = false; g_cache[i].last_use = ++g_cache_tick; }
                        else g_cache.erase(g_cache.begin() + i);   // a failed call leaves the plan in an unknown state: drop it
                        break;
                    }
            } else if (ok && plan_cache_capacity() > 0) {
                if (g_cache.size() >= plan_cache_capacity()) {   // evict the least recently used idle plan
                    size_t victim = g_cache.size();
                    for (size_t i = 0; i < g_cache.size(); ++i)
                        if (!g_cache[i].busy && (victim == g_cache.size() || g_cache[i].last_use < g_cache[victim].last_use)) victim = i;
                    if (victim < g_cache.size()) g_cache.erase(g_cache.begin() + victim);
                }
                if (g_cache.size() < plan_cache_capacity()) {
                    CachedPlan c;
                    c.plan = std::move(fresh);
                    c.desc = desc; c.device = dev; c.busy = false; c.last_use = ++g_cache_tick;
                    g_cache.push_back(std::move(c));
                }
            }
        };
        try {
            auto req = [](int rc) { if (rc != EMAGLS_OK) throw Error(rc, g_last_error); };
            if (desc.custom_basis) {
                req(emagls_plan_set_basis(p, Y_hrir, Y_mic));
            } else {
                req(emagls_plan_set_hrir_grid(p, azi, zen));
                if (mic_azi) req(emagls_plan_set_mic_grid(p, mic_azi, mic_zen));
            }
            req(emagls_plan_set_hrirs(p, hL, hR));
            if (atf) req(emagls_plan_set_atfs(p, atf, atf_azi, atf_zen));
            plan_execute(*p);
            req(emagls_plan_get_filters(p, wL, wR));
            if (mean_dev) {
                emagls_plan_info info;
                req(emagls_plan_get_info(p, &info));
                *mean_dev = info.mean_grid_dev_deg;
            }
        } catch (...) {
            release(false);
            throw;
        }
        release(true);
    });
}

}  // namespace

// designs per batch: 8 by default (one per XCD in the resident sweep), up to 16 (two per XCD) after emagls_set_batch_max / EMAGLS_BATCH_MAX
namespace {
std::atomic<int> g_batch_max{[] { const char* e = getenv("EMAGLS_BATCH_MAX"); return e ? std::max(1, std::min(REG_SWEEP_MAX, atoi(e))) : 8; }()};
thread_local int g_batch_max_override = 0;   // emagls_design_hrir_sets builds batches of 16 of its own whatever the caller's limit is
}
// work planes of the complex device-resident decode, grown on demand and kept (released by emagls_cache_clear)
namespace {
struct DecodeScratch {
    std::mutex mu;
    int device = -1;
    size_t cap_sig = 0, cap_w = 0, cap_tmp = 0;
    double *sig2 = nullptr, *w2L = nullptr, *w2R = nullptr, *tmp = nullptr;
    void release() {
        hipFree(sig2); hipFree(w2L); hipFree(w2R); hipFree(tmp);
        sig2 = w2L = w2R = tmp = nullptr; cap_sig = cap_w = cap_tmp = 0; device = -1;
    }
    void ensure(size_t nsig, size_t nw, size_t ntmp) {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        if (dev != device) { release(); device = dev; }
        if (nsig > cap_sig) { hipFree(sig2); sig2 = nullptr; cap_sig = 0; HIP_CHECK(hipMalloc(&sig2, sizeof(double) * nsig)); cap_sig = nsig; }
        if (nw > cap_w) {
            hipFree(w2L); hipFree(w2R); w2L = w2R = nullptr; cap_w = 0;
            HIP_CHECK(hipMalloc(&w2L, sizeof(double) * nw)); HIP_CHECK(hipMalloc(&w2R, sizeof(double) * nw)); cap_w = nw;
        }
        if (ntmp > cap_tmp) { hipFree(tmp); tmp = nullptr; cap_tmp = 0; HIP_CHECK(hipMalloc(&tmp, sizeof(double) * ntmp)); cap_tmp = ntmp; }
    }
};
DecodeScratch g_decode_scratch;
}  // namespace

// =============================================================================================
extern "C" {

const char* emagls_last_error(void) { return g_last_error.c_str(); }
int emagls_version(void) { return 100; }

int emagls_device_count(int* count) {
    return guarded([&] {
        if (!count) throw Error(EMAGLS_ERR_ARG, "null pointer");
        HIP_CHECK(hipGetDeviceCount(count));
    });
}
int emagls_set_device(int device) {
    return guarded([&] { HIP_CHECK(hipSetDevice(device)); });
}

void emagls_sets_cache_clear_internal();
void emagls_atfsets_cache_clear_internal();
void emagls_jobs_cache_clear_internal();
int emagls_cache_release_designs(void) {
    return guarded([&] {
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (size_t i = g_cache.size(); i-- > 0;)
                if (!g_cache[i].busy) g_cache.erase(g_cache.begin() + i);
        }
        emagls_sets_cache_clear_internal();
        emagls_atfsets_cache_clear_internal();
        emagls_jobs_cache_clear_internal();
    });
}
int emagls_cache_clear(void) {
    return guarded([&] {
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (size_t i = g_cache.size(); i-- > 0;)
                if (!g_cache[i].busy) g_cache.erase(g_cache.begin() + i);
        }
        decode_cache_clear();
        {
            std::lock_guard<std::mutex> lk(g_decode_scratch.mu);
            g_decode_scratch.release();
        }
        emagls_sets_cache_clear_internal();
        emagls_atfsets_cache_clear_internal();
        emagls_jobs_cache_clear_internal();
        BlockPool::get().clear();
    });
}

int emagls_fp64_peak_tflops(int which, double* tflops) {
    return guarded([&] {
        if (!tflops || which < 0 || which > 2) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        *tflops = measure_fp64_peak(which, 3);
    });
}

int emagls_self_test(int which, double* max_err) {
    return guarded([&] {
        if (!max_err || which < 0 || which > 2) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        *max_err = which == 0 ? reg_reduce_selftest() : gram_tile_selftest(which == 2);
    });
}

int emagls_fp64_peak_tflops_ex(int which, int burst, double* tflops, double* shader_mhz) {
    return guarded([&] {
        if (!tflops || which < 0 || which > 2) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        *tflops = measure_fp64_peak(which, 3, burst != 0, shader_mhz);
    });
}

int emagls_sh_basis(int order, int64_t ndirs, const double* azi, const double* zen, int basis, void* Y) {
    return guarded([&] {
        if (order < 0 || ndirs < 0 || !azi || !zen || !Y) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        if (ndirs == 0) return;
        const bool cb = basis == EMAGLS_BASIS_COMPLEX;
        const size_t S = (size_t)(order + 1) * (order + 1);
        double *d_azi = nullptr, *d_zen = nullptr, *d_tab = nullptr;
        void* d_Y = nullptr;
        auto cleanup = [&] { hipFree(d_azi); hipFree(d_zen); hipFree(d_tab); hipFree(d_Y); };
        try {
            HIP_CHECK(hipMalloc(&d_azi, sizeof(double) * ndirs));
            HIP_CHECK(hipMalloc(&d_zen, sizeof(double) * ndirs));
            HIP_CHECK(hipMalloc(&d_tab, sizeof(double) * sh_coeff_count(order)));
            HIP_CHECK(hipMalloc(&d_Y, esz(cb) * S * ndirs));
            HIP_CHECK(hipMemcpy(d_azi, azi, sizeof(double) * ndirs, hipMemcpyDefault));
            HIP_CHECK(hipMemcpy(d_zen, zen, sizeof(double) * ndirs, hipMemcpyDefault));
            launch_sh_coeff(order, d_tab, nullptr);
            launch_sh_basis(order, ndirs, d_azi, d_zen, d_tab, cb, d_Y, ndirs, nullptr);
            HIP_CHECK(hipDeviceSynchronize());
            HIP_CHECK(hipMemcpy(Y, d_Y, esz(cb) * S * ndirs, hipMemcpyDefault));
        } catch (...) { cleanup(); throw; }
        cleanup();
    });
}

int emagls_sh_basis_device(int order, int64_t ndirs, const double* d_azi, const double* d_zen, int basis, void* d_Y,
                           void* stream) {
    return guarded([&] {
        if (order < 0 || ndirs < 0 || !d_azi || !d_zen || !d_Y) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        if (ndirs == 0) return;
        // the recurrence table is tiny and depends only on the order: cached per (thread, order)
        thread_local int tab_order = -1;
        thread_local double* tab = nullptr;
        hipStream_t st = (hipStream_t)stream;
        if (tab_order != order) {
            if (tab) { HIP_CHECK(hipFree(tab)); tab = nullptr; }
            HIP_CHECK(hipMalloc(&tab, sizeof(double) * sh_coeff_count(order)));
            tab_order = order;
            launch_sh_coeff(order, tab, st);
        }
        launch_sh_basis(order, ndirs, d_azi, d_zen, tab, basis == EMAGLS_BASIS_COMPLEX, d_Y, ndirs, st);
    });
}

int emagls_modal_bn(int order, int64_t nfreq, const double* kr, void* bn) {
    return guarded([&] {
        if (order < 0 || nfreq < 0 || !kr || !bn) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        if (nfreq == 0) return;
        double* d_kr = nullptr;
        void* d_bn = nullptr;
        auto cleanup = [&] { hipFree(d_kr); hipFree(d_bn); };
        try {
            HIP_CHECK(hipMalloc(&d_kr, sizeof(double) * nfreq));
            HIP_CHECK(hipMalloc(&d_bn, sizeof(cplx) * nfreq * (order + 1)));
            HIP_CHECK(hipMemcpy(d_kr, kr, sizeof(double) * nfreq, hipMemcpyDefault));
            launch_modal_bn(order, nfreq, d_kr, 1.0, 1.0, d_bn, 1, nfreq, nullptr);  // column-major [nfreq x (order+1)]
            HIP_CHECK(hipDeviceSynchronize());
            HIP_CHECK(hipMemcpy(bn, d_bn, sizeof(cplx) * nfreq * (order + 1), hipMemcpyDefault));
        } catch (...) { cleanup(); throw; }
        cleanup();
    });
}

// ---------------------------------------------------------------------------------------------
int emagls_plan_create(const emagls_design_desc* desc, emagls_plan** plan) {
    return guarded([&] {
        if (!desc || !plan) throw Error(EMAGLS_ERR_ARG, "null pointer");
        std::unique_ptr<emagls_plan> p(new emagls_plan);
        p->d = *desc;
        plan_setup(*p);
        *plan = p.release();
    });
}
int emagls_plan_destroy(emagls_plan* plan) {
    return guarded([&] {
        DeviceGuard dg(plan ? plan->device : -1);
        delete plan;
    });
}
int emagls_plan_set_hrir_grid(emagls_plan* p, const double* azi, const double* zen) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p || !azi) throw Error(EMAGLS_ERR_ARG, "null pointer");
        // a horizontal HRIR set (getMagLsFilters2D) has no zenith argument: pi/2 for every direction
        std::vector<double> equator;
        if (p->d.kind == EMAGLS_KIND_MAGLS_2D) { equator.assign((size_t)p->d.ndirs, kPi / 2.0); zen = equator.data(); }
        if (!zen) throw Error(EMAGLS_ERR_ARG, "null pointer");
        p->upload("hrir_azi", azi, sizeof(double) * p->d.ndirs);
        p->upload("hrir_zen", zen, sizeof(double) * p->d.ndirs);
        HIP_CHECK(hipStreamSynchronize(p->stream));
        p->have_hrir_grid = true;
        ++p->atf_side_version;
    });
}
int emagls_plan_set_mic_grid(emagls_plan* p, const double* azi, const double* zen) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p || !azi) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (!p->has("mic_azi")) throw Error(EMAGLS_ERR_ARG, "this design kind has no microphone grid");
        // an equatorial array (EMAinCH) has no zenith argument: pi/2 for every microphone (getEMagLsFiltersEMAinCH.m:60)
        std::vector<double> equator;
        if (p->d.kind == EMAGLS_KIND_EMA_CH || p->d.kind == EMAGLS_KIND_EMA_SH) { equator.assign((size_t)p->d.nmics, kPi / 2.0); zen = equator.data(); }
        if (!zen) throw Error(EMAGLS_ERR_ARG, "null pointer");
        p->upload("mic_azi", azi, sizeof(double) * p->d.nmics);
        p->upload("mic_zen", zen, sizeof(double) * p->d.nmics);
        if (p->has("smap")) {   // the synthesising sweep's row order of the microphones: antipodal pairs first (sweep_synth.hip)
            int smap[34];
            synth_pairing(azi, zen, (int)p->d.nmics, smap);
            p->upload("smap", smap, sizeof smap);
            p->synth_units = smap[32] + smap[33];
            HIP_CHECK(hipStreamSynchronize(p->stream));   // (the host array goes out of scope)
        }
        // kr = 2*pi*f/C * smaRadius on f = linspace(0, fs/2, P)   (getSMAIRMatrix.m:90,107)
        std::vector<double> kr(p->P);
        for (int k = 0; k < p->P; ++k) {
            const double f = (double)k * (p->d.fs / 2.0) / (double)(p->P - 1);
            kr[k] = 2.0 * kPi * f / C_SOUND * p->d.mic_radius;
        }
        p->upload("kr", kr.data(), sizeof(double) * p->P);
        HIP_CHECK(hipStreamSynchronize(p->stream));
        p->have_mic_grid = true;
        ++p->atf_side_version;   // (a geometry-sharing batch compares the grids again)
    });
}
int emagls_plan_set_basis(emagls_plan* p, const void* Y_hrir, const void* Y_mic) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p || !Y_hrir) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (!p->custom_basis) throw Error(EMAGLS_ERR_ARG, "the plan was not created with custom_basis = 1");
        const size_t es = esz(p->cplx_basis);
        // MATLAB layout [ndirs x S] column-major -> Ycm [S][ldD]
        HIP_CHECK(hipMemcpy2DAsync(p->get("Ycm"), (size_t)p->ldD * es, Y_hrir, (size_t)p->D * es, (size_t)p->D * es, (size_t)p->S, hipMemcpyDefault,
                                   p->stream));
        if (array_kind(p->d.kind)) {
            if (!Y_mic) throw Error(EMAGLS_ERR_ARG, "the array designs need the SH matrix of the microphone grid too");
            p->upload("Ymic_cm", Y_mic, es * (size_t)p->S * p->d.nmics);   // [nmics x S] column-major == [S][M]
            std::vector<double> kr(p->P);
            for (int k = 0; k < p->P; ++k) kr[k] = 2.0 * kPi * ((double)k * (p->d.fs / 2.0) / (double)(p->P - 1)) / C_SOUND * p->d.mic_radius;
            p->upload("kr", kr.data(), sizeof(double) * p->P);
        }
        HIP_CHECK(hipStreamSynchronize(p->stream));
        p->have_basis = true;
    });
}
int emagls_plan_set_hrirs(emagls_plan* p, const double* hL, const double* hR) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p || !hL || !hR) throw Error(EMAGLS_ERR_ARG, "null pointer");
        p->upload("hL", hL, sizeof(double) * p->d.nsamp * p->d.ndirs);
        p->upload("hR", hR, sizeof(double) * p->d.nsamp * p->d.ndirs);
        HIP_CHECK(hipStreamSynchronize(p->stream));
        p->have_hrirs = true;
    });
}
int emagls_plan_set_atfs(emagls_plan* p, const double* atf, const double* azi, const double* zen) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p || !atf || !azi || !zen) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (!p->has("atf")) throw Error(EMAGLS_ERR_ARG, "this design kind has no ATFs");
        p->upload("atf", atf, sizeof(double) * p->d.atf_taps * p->d.nmics * p->d.natf);
        p->upload("atf_azi", azi, sizeof(double) * p->d.natf);
        p->upload("atf_zen", zen, sizeof(double) * p->d.natf);
        HIP_CHECK(hipStreamSynchronize(p->stream));
        p->have_atfs = true;
        ++p->atf_side_version;
    });
}
int emagls_plan_execute(emagls_plan* p) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p) throw Error(EMAGLS_ERR_ARG, "null pointer");
        plan_execute(*p);
    });
}
int emagls_plan_synchronize(emagls_plan* p) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p) throw Error(EMAGLS_ERR_ARG, "null pointer");
        HIP_CHECK(hipStreamSynchronize(p->sync_stream ? p->sync_stream : p->stream));
    });
}
int emagls_plan_get_filters(emagls_plan* p, void* wL, void* wR) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p || !wL || !wR) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (!p->executed) throw Error(EMAGLS_ERR_ARG, "plan has not been executed");
        HIP_CHECK(hipStreamSynchronize(p->sync_stream ? p->sync_stream : p->stream));
        plan_check_flags(*p);
        const size_t bytes = (p->out_cplx ? sizeof(cplx) : sizeof(double)) * (size_t)p->out_rows * p->out_cols;
        HIP_CHECK(hipMemcpy(wL, p->get("wL"), bytes, hipMemcpyDefault));
        HIP_CHECK(hipMemcpy(wR, p->get("wR"), bytes, hipMemcpyDefault));
    });
}
int emagls_plan_sweep_form_in_batch(emagls_plan* p, int designs, int* form) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p || !form || designs < 1 || designs > REG_SWEEP_MAX) throw Error(EMAGLS_ERR_ARG, "emagls_plan_sweep_form_in_batch: null pointer or designs outside 1..32");
        std::vector<emagls_plan*> same((size_t)designs, p);
        const bool reg = p->synth && reg_sweep_wanted(same.data(), designs);
        *form = p->d.kind == EMAGLS_KIND_LS ? 0 : (p->synth ? (reg ? 3 : 2) : (p->sweep_persist ? 1 : 0));
    });
}
int emagls_plan_get_info(emagls_plan* p, emagls_plan_info* info) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p || !info) throw Error(EMAGLS_ERR_ARG, "null pointer");
        std::memset(info, 0, sizeof *info);
        info->nfft = p->nfft; info->num_pos_freqs = p->P; info->k_cut = p->k_cut; info->sim_order = p->simOrder;
        info->num_sh_sim = p->S; info->num_channels = p->C; info->out_is_complex = p->out_cplx;
        info->out_rows = p->out_rows; info->out_cols = p->out_cols; info->num_sweep_launches = p->sweep_launches;
        info->device_bytes = p->total_bytes;
        info->gram_from = p->gram_from; info->hh_end = p->hh_end; info->hh_orders = p->n_h + 1; info->g_first = p->g0;
        info->sim_order_own = array_kind(p->d.kind) ? p->simOrderOwn : p->simOrder;
        bool reg = false;   // the form the next sweep takes (decided per launch: a batch's for its members)
        if (p->synth) {
            bool whole = p->owner != nullptr;
            if (whole) for (auto* q : p->owner->plans) whole = whole && q != nullptr;
            reg = whole ? reg_sweep_wanted(p->owner->plans.data(), (int)p->owner->plans.size()) : reg_sweep_wanted(&p, 1);
        }
        info->sweep_form = p->d.kind == EMAGLS_KIND_LS ? 0 : (p->synth ? (reg ? 3 : 2) : (p->sweep_persist ? 1 : 0));
        info->sweep_units = p->synth ? p->synth_units : 0;
        if (p->executed) {
            HIP_CHECK(hipStreamSynchronize(p->stream));
            double g[2];
            HIP_CHECK(hipMemcpy(g, p->get("grpd"), sizeof g, hipMemcpyDeviceToHost));
            info->grp_delay_l = g[0]; info->grp_delay_r = g[1];
            if (p->has("mean_dev")) HIP_CHECK(hipMemcpy(&info->mean_grid_dev_deg, p->get("mean_dev"), sizeof(double), hipMemcpyDeviceToHost));
        }
    });
}
int emagls_plan_set_profiling(emagls_plan* p, int level) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p) throw Error(EMAGLS_ERR_ARG, "null pointer");
        p->prof_level = level;
    });
}
int emagls_plan_set_streams(emagls_plan* p, int nstreams) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (nstreams < 1 || nstreams > 4) throw Error(EMAGLS_ERR_ARG, "nstreams must be 1..4");
        HIP_CHECK(hipStreamSynchronize(p->stream));
        if (p->graph_exec) { HIP_CHECK(hipGraphExecDestroy(p->graph_exec)); p->graph_exec = nullptr; }
        if (p->graph) { HIP_CHECK(hipGraphDestroy(p->graph)); p->graph = nullptr; }
        if (p->pre_exec) { HIP_CHECK(hipGraphExecDestroy(p->pre_exec)); p->pre_exec = nullptr; }
        if (p->pre_graph) { HIP_CHECK(hipGraphDestroy(p->pre_graph)); p->pre_graph = nullptr; }
        p->eager_runs = 0;   // (the next execute runs eagerly again, the one after it captures with the new stream count)
        p->nstreams = nstreams;
    });
}
int emagls_plan_num_stages(emagls_plan* p) { return p ? (int)p->stage_names.size() : 0; }
const char* emagls_plan_stage_name(emagls_plan* p, int i) {
    if (!p || i < 0 || i >= (int)p->stage_names.size()) return "";
    return p->stage_names[i].c_str();
}
int emagls_plan_stage_times(emagls_plan* p, double* ms, int n) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p || !ms) throw Error(EMAGLS_ERR_ARG, "null pointer");
        HIP_CHECK(hipStreamSynchronize(p->stream));
        const int ns = (int)p->stage_names.size();
        for (int i = 0; i < n; ++i) {
            ms[i] = 0.0;
            if (i >= 1 && i < ns) {
                float t = 0.f;
                HIP_CHECK(hipEventElapsedTime(&t, p->stage_events[i - 1], p->stage_events[i]));
                ms[i] = t;
            }
        }
    });
}
int emagls_plan_sweep_kernel_time(emagls_plan* p, double* total_ms, int* launches) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p || !total_ms || !launches) throw Error(EMAGLS_ERR_ARG, "null pointer");
        HIP_CHECK(hipStreamSynchronize(p->stream));
        double tot = 0.0;
        int n = 0;
        if (p->prof_level >= 2) {
            for (int i = 0; i < p->sweep_launches && 2 * (size_t)i + 1 < p->sweep_events.size(); ++i) {
                float t = 0.f;
                HIP_CHECK(hipEventElapsedTime(&t, p->sweep_events[2 * i], p->sweep_events[2 * i + 1]));
                tot += t;
                ++n;
            }
        }
        *total_ms = tot;
        *launches = n;
    });
}
int emagls_plan_debug_buffer(emagls_plan* p, const char* name, void* dst, size_t* nbytes) {
    return guarded([&] {
        DeviceGuard dg(p ? p->device : -1);
        if (!p || !name || !nbytes) throw Error(EMAGLS_ERR_ARG, "null pointer");
        auto it = p->bufs.find(name);
        if (it == p->bufs.end()) throw Error(EMAGLS_ERR_ARG, std::string("unknown buffer ") + name);
        if (!dst) { *nbytes = it->second.bytes; return; }
        HIP_CHECK(hipStreamSynchronize(p->stream));
        const size_t n = std::min(*nbytes, it->second.bytes);
        HIP_CHECK(hipMemcpy(dst, it->second.p, n, hipMemcpyDeviceToHost));
        *nbytes = n;
    });
}
void* emagls_plan_stream(emagls_plan* p) { return p ? (void*)p->stream : nullptr; }

int emagls_batch_create(emagls_plan** plans, int nplans, emagls_batch** batch) {
    return guarded([&] {
        if (!plans || !batch || nplans < 1) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        // Up to 8 designs per batch by default.  9 to 16 share one sweep launch with two workgroups per CU, which covers EVERY
        // CU: safe and fastest when nothing else runs on the device, but another stream's kernel that needs a whole CU
        // (e.g. factor_qr: 1024 threads x 128 registers) makes the dispatcher hold back the sweep's remaining workgroups
        // while the resident ones wait for them -- observed as a 0.4 s stall until the sweep's own time-out falls back to the
        // launch-per-bin form.  Hence opt-in: EMAGLS_BATCH_MAX=16.
        const int batch_max = std::max(g_batch_max.load(), g_batch_max_override);
        if (nplans > batch_max)
            throw Error(EMAGLS_ERR_UNSUPPORTED, batch_max >= REG_SWEEP_MAX ? "at most 32 designs per batch"
                                                                             : "at most 8 designs per batch (emagls_set_batch_max(16) / EMAGLS_BATCH_MAX=16 allows 16)");
        std::unique_ptr<emagls_batch> b(new emagls_batch);
        for (int j = 0; j < nplans; ++j) {
            emagls_plan* p = plans[j];
            if (!p) throw Error(EMAGLS_ERR_ARG, "null plan");
            if (!array_kind(p->d.kind) && p->d.kind != EMAGLS_KIND_FROM_ATF && !magls_kind(p->d.kind) && p->d.kind != EMAGLS_KIND_LS)
                throw Error(EMAGLS_ERR_UNSUPPORTED, "batches take eMagLS / eMagLS2 / EMAinCH / EMAinSH plans, LS / MagLS / MagLS-2D plans, or FromAtf plans "
                                                    "(subjects of one ATF set)");
            if ((p->d.kind == EMAGLS_KIND_LS) != (plans[0]->d.kind == EMAGLS_KIND_LS) ||
                (p->d.kind == EMAGLS_KIND_LS && (p->cplx_basis != plans[0]->cplx_basis || p->d.order != plans[0]->d.order || p->d.nsamp != plans[0]->d.nsamp)))
                throw Error(EMAGLS_ERR_ARG, "LS plans share a batch only with LS plans of the same order, basis and HRIR length");
            if ((p->d.kind == EMAGLS_KIND_FROM_ATF) != (plans[0]->d.kind == EMAGLS_KIND_FROM_ATF))
                throw Error(EMAGLS_ERR_ARG, "FromAtf plans cannot share a batch with array designs");
            if (magls_kind(p->d.kind) != magls_kind(plans[0]->d.kind) || (magls_kind(p->d.kind) && p->d.kind != plans[0]->d.kind))
                throw Error(EMAGLS_ERR_ARG, "MagLS plans share a batch only with MagLS plans of the same kind");
            if (magls_kind(p->d.kind) && (p->diffuse != plans[0]->diffuse || p->cplx_basis != plans[0]->cplx_basis || p->d.order != plans[0]->d.order))
                throw Error(EMAGLS_ERR_ARG, "all designs of a batch must have the same shape (order, basis, constraint)");
            if (p->wide) throw Error(EMAGLS_ERR_UNSUPPORTED, "designs with more than 32 channels run one at a time");
            if (p->owner) throw Error(EMAGLS_ERR_ARG, "a plan belongs to another batch (destroy that batch first)");
            if (p->device != plans[0]->device) throw Error(EMAGLS_ERR_ARG, "the plans of a batch must live on one device");
            for (int i = 0; i < j; ++i) if (plans[i] == p) throw Error(EMAGLS_ERR_ARG, "the same plan appears twice in the batch");
            const emagls_plan* q = plans[0];
            if (p->P != q->P || p->kcut0 != q->kcut0 || p->C != q->C || p->nWG_dense != q->nWG_dense || p->D != q->D || p->sweep_persist != q->sweep_persist ||
                p->Dm != q->Dm || p->d.natf != q->d.natf || p->d.atf_taps != q->d.atf_taps || p->d.len != q->d.len)
                throw Error(EMAGLS_ERR_ARG, "all designs of a batch must have the same shape (directions, channels, bins, k_cut)");
            b->plans.push_back(p);
        }
        b->device = b->plans[0]->device;
        DeviceGuard dg(b->device);
        b->stream = StreamPool::get().take();
        if (const char* ng = getenv("EMAGLS_NO_GRAPH")) b->use_graph = !(ng[0] == '1');
        // one persistent sweep launch keeps designs x nWG workgroups resident: one per CU up to 8 designs (one design per XCD),
        // two per CU beyond (77 KB of LDS and 5 waves per workgroup)
        // (up to 8 designs: 16 CUs stay free of sweep workgroups, so that kernels of other batches which need a whole CU keep
        // making progress and the dispatcher never has a reason to hold the sweep's own workgroups back)
        // (decided here, before any launch, from the runtime's occupancy of the kernel variant: persist_sweep_fits)
        const bool array_batch = b->plans[0]->d.kind != EMAGLS_KIND_FROM_ATF && !magls_kind(b->plans[0]->d.kind) && b->plans[0]->d.kind != EMAGLS_KIND_LS;
        for (auto* p : b->plans) HIP_CHECK(hipStreamSynchronize(p->stream));
        // (the sweep's form first -- one launch serves every design: the synthesising forms only when all qualify --, then its residency)
        trace_mark("batch create: plans synchronised");
        if (array_batch) batch_unify_synth(*b);
        trace_mark("batch create: sweep form unified");
        batch_decide_residency(*b);
        trace_mark("batch create: residency decided");
        if (nplans > SWEEP_MULTI_MAX && !(b->plans[0]->synth && b->plans[0]->sweep_persist && reg_sweep_wanted(b->plans.data(), nplans)))
            throw Error(EMAGLS_ERR_UNSUPPORTED, "more than 16 designs per batch: only array designs that take the register-resident sweep (built-in SH basis, "
                                                "microphone grids set, at most 18 antipodal pairs + single microphones, a launch the device can hold)");
        for (auto* p : b->plans) {
            p->nstreams = 1;
            p->prof_level = 0;
            p->sync_stream = b->stream;
            p->owner = b.get();
        }
        b->atf = b->plans[0]->d.kind == EMAGLS_KIND_FROM_ATF;
        b->magls = magls_kind(b->plans[0]->d.kind) || b->plans[0]->d.kind == EMAGLS_KIND_LS;
        if (!b->atf && !b->magls) { batch_unify_synth(*b); batch_try_lanes(*b); }
        trace_mark("batch create: lanes tried");
        *batch = b.release();
    });
}
int emagls_batch_execute(emagls_batch* b) {
    return guarded([&] {
        DeviceGuard dg(b ? b->device : -1);
        if (!b) throw Error(EMAGLS_ERR_ARG, "null pointer");
        batch_execute(*b);
    });
}
int emagls_batch_synchronize(emagls_batch* b) {
    return guarded([&] {
        DeviceGuard dg(b ? b->device : -1);
        if (!b) throw Error(EMAGLS_ERR_ARG, "null pointer");
        HIP_CHECK(hipStreamSynchronize(b->stream));
    });
}
int emagls_batch_get_filters(emagls_batch* b, void* const* wL, void* const* wR) {
    return guarded([&] {
        DeviceGuard dg(b ? b->device : -1);
        if (!b || !wL || !wR) throw Error(EMAGLS_ERR_ARG, "null pointer");
        const size_t n = b->plans.size();
        for (size_t j = 0; j < n; ++j) {
            if (!wL[j] || !wR[j]) throw Error(EMAGLS_ERR_ARG, "null pointer");
            if (!b->plans[j]) throw Error(EMAGLS_ERR_ARG, "a plan of this batch has been destroyed");
            if (!b->plans[j]->executed) throw Error(EMAGLS_ERR_ARG, "batch has not been executed");
        }
        emagls_plan& p0 = *b->plans[0];
        const size_t bytes = (p0.out_cplx ? sizeof(cplx) : sizeof(double)) * (size_t)p0.out_rows * p0.out_cols;
        // the copies are ordered behind the batch on its stream; one synchronisation for everything
        std::vector<int> flags(NFLAG * n, 0);
        for (int attempt = 0; attempt < 3; ++attempt) {
            if (b->lanes)
                HIP_CHECK(hipMemcpy2DAsync(flags.data(), NFLAG * sizeof(int), p0.get("flag"), b->stride, NFLAG * sizeof(int), n,
                                           hipMemcpyDeviceToHost, b->stream));
            else
                for (size_t j = 0; j < n; ++j)
                    HIP_CHECK(hipMemcpyAsync(&flags[NFLAG * j], b->plans[j]->get("flag"), NFLAG * sizeof(int), hipMemcpyDeviceToHost, b->stream));
            // lane batch into device buffers of this GPU: one scatter launch instead of 2 n copies (0.25 ms on the stream for 16
            // designs -- the tail of a short run's timed region)
            bool scattered = false;
            if (b->lanes && n <= 32 && bytes % 16 == 0) {
                bool dev_dst = true;
                for (size_t j = 0; j < n && dev_dst; ++j)
                    for (void* q : {wL[j], wR[j]}) {
                        hipPointerAttribute_t at{};
                        if (hipPointerGetAttributes(&at, q) != hipSuccess) { (void)hipGetLastError(); dev_dst = false; break; }
                        if (at.type != hipMemoryTypeDevice || at.device != b->device || ((uintptr_t)q & 15)) { dev_dst = false; break; }
                    }
                if (dev_dst) {
                    LanePtrs lp{};
                    for (size_t j = 0; j < n; ++j) { lp.p[2 * j] = wL[j]; lp.p[2 * j + 1] = wR[j]; }
                    launch_scatter_lanes(p0.get("wL"), p0.get("wR"), b->stride, bytes, (int)n, lp, b->stream);
                    scattered = true;
                }
            }
            if (!scattered)
                for (size_t j = 0; j < n; ++j) {
                    HIP_CHECK(hipMemcpyAsync(wL[j], b->plans[j]->get("wL"), bytes, hipMemcpyDefault, b->stream));
                    HIP_CHECK(hipMemcpyAsync(wR[j], b->plans[j]->get("wR"), bytes, hipMemcpyDefault, b->stream));
                }
            HIP_CHECK(hipStreamSynchronize(b->stream));
            bool redo = false;
            for (size_t j = 0; j < n; ++j) redo = plan_recover(*b->plans[j], &flags[NFLAG * j], false) || redo;
            if (!redo) break;
            // recoverable: a Gram-route bin worse conditioned than estimated, or a persistent sweep that did not become resident
            batch_redo(*b, flags);
        }
        for (auto* q : b->plans)   // (a MagLS re-run on the launch-per-bin sweeps is done: the next execute starts on the persistent form again)
            if (q->persist_suspended) { q->persist_suspended = false; q->sweep_persist = true; }
        for (size_t j = 0; j < n; ++j) throw_fatal_flags(&flags[NFLAG * j]);
    });
}
int emagls_batch_lane_mode(emagls_batch* b, int* lanes) {
    return guarded([&] {
        DeviceGuard dg(b ? b->device : -1);
        if (!b || !lanes) throw Error(EMAGLS_ERR_ARG, "null pointer");
        *lanes = b->lanes ? 1 : 0;
    });
}
int emagls_set_batch_max(int max_designs, int* previous) {
    return guarded([&] {
        if (max_designs < 1 || max_designs > REG_SWEEP_MAX) throw Error(EMAGLS_ERR_ARG, "a batch holds 1..32 designs (more than 16: array designs on the register-resident sweep)");
        const int prev = g_batch_max.exchange(max_designs);
        if (previous) *previous = prev;
    });
}
int emagls_batch_set_geometry_sharing(emagls_batch* b, int enable) {
    return guarded([&] {
        DeviceGuard dg(b ? b->device : -1);
        if (!b) throw Error(EMAGLS_ERR_ARG, "null pointer");
        HIP_CHECK(hipStreamSynchronize(b->stream));
        b->geo_want = enable != 0;
        b->geo_checked_version = ~0ull;
    });
}
int emagls_batch_shares_geometry(emagls_batch* b, int* shared) {
    return guarded([&] {
        if (!b || !shared) throw Error(EMAGLS_ERR_ARG, "null pointer");
        *shared = b->geo_share ? 1 : 0;
    });
}
int emagls_batch_shares_atf_side(emagls_batch* b, int* shared) {
    return guarded([&] {
        if (!b || !shared) throw Error(EMAGLS_ERR_ARG, "null pointer");
        *shared = b->atf_share ? 1 : 0;
    });
}
int emagls_batch_set_stream(emagls_batch* b, void* stream) {
    return guarded([&] {
        DeviceGuard dg(b ? b->device : -1);
        if (!b || !stream) throw Error(EMAGLS_ERR_ARG, "null pointer");
        HIP_CHECK(hipStreamSynchronize(b->stream));
        if (b->own_stream) emagls::pool_stream_give(b->stream);
        b->stream = (hipStream_t)stream;      // (captured graphs are not tied to a stream: they replay on the new one)
        b->own_stream = false;
        for (auto* p : b->plans) if (p) p->sync_stream = b->stream;
    });
}
int emagls_batch_set_streams(emagls_batch* b, int nstreams) {
    return guarded([&] {
        DeviceGuard dg(b ? b->device : -1);
        if (!b) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (nstreams < 1 || nstreams > 4) throw Error(EMAGLS_ERR_ARG, "nstreams must be 1..4");
        HIP_CHECK(hipStreamSynchronize(b->stream));
        for (int i = 0; i < nstreams - 1; ++i) if (!b->side[i]) b->side[i] = emagls::pool_stream_take();
        if (nstreams != b->nstreams) drop_batch_graphs(*b);   // (the next execute runs eagerly, the one after it captures the forks)
        b->nstreams = nstreams;
    });
}
int emagls_batch_set_stage_order(emagls_batch* b, int order) {
    return guarded([&] {
        DeviceGuard dg(b ? b->device : -1);
        if (!b) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (order < 0 || order > 2) throw Error(EMAGLS_ERR_ARG, "stage order must be 0, 1 or 2");
        HIP_CHECK(hipStreamSynchronize(b->stream));
        if (order != b->order_hint) drop_batch_graphs(*b);
        b->order_hint = order;
    });
}
int emagls_batch_set_side_stream(emagls_batch* b, void* stream) {
    return guarded([&] {
        DeviceGuard dg(b ? b->device : -1);
        if (!b || !stream) throw Error(EMAGLS_ERR_ARG, "null pointer");
        HIP_CHECK(hipStreamSynchronize(b->stream));
        if (b->side[0] && !b->side0_external) { HIP_CHECK(hipStreamSynchronize(b->side[0])); emagls::pool_stream_give(b->side[0]); }
        b->side[0] = (hipStream_t)stream;
        b->side0_external = true;
    });
}
int emagls_batch_set_profiling(emagls_batch* b, int level) {
    return guarded([&] {
        DeviceGuard dg(b ? b->device : -1);
        if (!b) throw Error(EMAGLS_ERR_ARG, "null pointer");
        HIP_CHECK(hipStreamSynchronize(b->stream));
        if (level >= 1)
            for (auto& e : b->sweep_ev) if (!e) HIP_CHECK(hipEventCreate(&e));
        b->prof_level = level;
    });
}
int emagls_batch_sweep_time(emagls_batch* b, double* ms) {
    return guarded([&] {
        DeviceGuard dg(b ? b->device : -1);
        if (!b || !ms) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (b->prof_level < 1 || !b->sweep_ev[0]) throw Error(EMAGLS_ERR_ARG, "batch profiling is off");
        HIP_CHECK(hipStreamSynchronize(b->stream));
        float t = 0.f;
        HIP_CHECK(hipEventElapsedTime(&t, b->sweep_ev[0], b->sweep_ev[1]));
        *ms = t;
    });
}
int emagls_batch_destroy(emagls_batch* b) {
    return guarded([&] {
        DeviceGuard dg(b ? b->device : -1);
        delete b;
    });
}

// ---------------------------------------------------------------------------------------------
int emagls_get_ls_filters(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi,
                          const double* zen, int order, int basis, void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_LS; d.basis = basis; d.order = order; d.nsamp = nsamp; d.ndirs = ndirs;
    return one_shot(d, hL, hR, azi, zen, nullptr, nullptr, nullptr, nullptr, nullptr, wL, wR, nullptr);
}
int emagls_get_magls_filters(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi,
                             const double* zen, int order, double fs, int64_t len, int basis, void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_MAGLS; d.basis = basis; d.order = order; d.fs = fs; d.len = len; d.nsamp = nsamp; d.ndirs = ndirs;
    return one_shot(d, hL, hR, azi, zen, nullptr, nullptr, nullptr, nullptr, nullptr, wL, wR, nullptr);
}
int emagls_get_emagls_filters(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi,
                              const double* zen, double mic_radius, const double* mic_azi, const double* mic_zen,
                              int64_t nmics, int order, double fs, int64_t len, int basis, void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_EMAGLS; d.basis = basis; d.order = order; d.fs = fs; d.len = len; d.nsamp = nsamp; d.ndirs = ndirs;
    d.mic_radius = mic_radius; d.nmics = nmics;
    if (!mic_azi || !mic_zen) { g_last_error = "null microphone grid"; return EMAGLS_ERR_ARG; }
    return one_shot(d, hL, hR, azi, zen, mic_azi, mic_zen, nullptr, nullptr, nullptr, wL, wR, nullptr);
}
int emagls_get_emagls2_filters(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi,
                               const double* zen, double mic_radius, const double* mic_azi, const double* mic_zen,
                               int64_t nmics, int order, double fs, int64_t len, int basis, void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_EMAGLS2; d.basis = basis; d.order = order; d.fs = fs; d.len = len; d.nsamp = nsamp; d.ndirs = ndirs;
    d.mic_radius = mic_radius; d.nmics = nmics;
    if (!mic_azi || !mic_zen) { g_last_error = "null microphone grid"; return EMAGLS_ERR_ARG; }
    return one_shot(d, hL, hR, azi, zen, mic_azi, mic_zen, nullptr, nullptr, nullptr, wL, wR, nullptr);
}
int emagls_get_emagls_filters_ema_in_ch(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi,
                                        const double* zen, double mic_radius, const double* mic_azi, int64_t nmics, int order,
                                        double fs, int64_t len, int basis, void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_EMA_CH; d.basis = basis; d.order = order; d.fs = fs; d.len = len; d.nsamp = nsamp; d.ndirs = ndirs;
    d.mic_radius = mic_radius; d.nmics = nmics;
    if (!mic_azi || nmics < 1) { g_last_error = "invalid array geometry"; return EMAGLS_ERR_ARG; }
    std::vector<double> mic_zen((size_t)nmics, kPi / 2.0);   // getEMagLsFiltersEMAinCH.m:60: equatorial array
    return one_shot(d, hL, hR, azi, zen, mic_azi, mic_zen.data(), nullptr, nullptr, nullptr, wL, wR, nullptr);
}
int emagls_get_emagls_filters_ema_in_sh(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi,
                                        const double* zen, double mic_radius, const double* mic_azi, int64_t nmics, int order,
                                        double fs, int64_t len, int basis, void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_EMA_SH; d.basis = basis; d.order = order; d.fs = fs; d.len = len; d.nsamp = nsamp; d.ndirs = ndirs;
    d.mic_radius = mic_radius; d.nmics = nmics;
    if (!mic_azi || nmics < 1) { g_last_error = "invalid array geometry"; return EMAGLS_ERR_ARG; }
    std::vector<double> mic_zen((size_t)nmics, kPi / 2.0);   // getEMagLsFiltersEMAinSH.m:58: equatorial array
    return one_shot(d, hL, hR, azi, zen, mic_azi, mic_zen.data(), nullptr, nullptr, nullptr, wL, wR, nullptr);
}
int emagls_get_emagls_filters_from_atf(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs,
                                       const double* azi, const double* zen, const double* atf_irs, int64_t atf_taps,
                                       int64_t nmics, int64_t natf, const double* atf_azi, const double* atf_zen, double fs,
                                       int64_t filter_len, double f_trans, double* wL, double* wR, double* mean_dev) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_FROM_ATF; d.basis = EMAGLS_BASIS_REAL; d.fs = fs; d.len = filter_len; d.nsamp = nsamp; d.ndirs = ndirs;
    d.nmics = nmics; d.f_trans = f_trans; d.atf_taps = atf_taps; d.natf = natf;
    if (!atf_irs || !atf_azi || !atf_zen) { g_last_error = "null ATF set"; return EMAGLS_ERR_ARG; }
    return one_shot(d, hL, hR, azi, zen, nullptr, nullptr, atf_irs, atf_azi, atf_zen, wL, wR, mean_dev);
}

// common body of the two decode entry points
static int decode_entry(const void* in, bool in_cplx, int64_t nsamp, int64_t nch, const void* wL, const void* wR, bool w_cplx, int64_t len,
                        int compensate_delay, double* out, double* imag_abs_sum) {
    return guarded([&] {
        if (!in || !wL || !wR || !out) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (nsamp < 0 || nch < 1 || len < 1) throw Error(EMAGLS_ERR_ARG, "invalid shape");
        if (imag_abs_sum) imag_abs_sum[0] = imag_abs_sum[1] = 0.0;
        if (nsamp == 0) return;
        const bool any_cplx = in_cplx || w_cplx;
        const size_t es_in = in_cplx ? sizeof(cplx) : sizeof(double), es_w = w_cplx ? sizeof(cplx) : sizeof(double);
        void *d_in = nullptr, *d_wL = nullptr, *d_wR = nullptr;
        double *d_out = nullptr, *d_sig2 = nullptr, *d_w2L = nullptr, *d_w2R = nullptr, *d_tmp = nullptr;
        hipStream_t st = nullptr;
        auto cleanup = [&] {
            hipFree(d_in); hipFree(d_wL); hipFree(d_wR); hipFree(d_out); hipFree(d_sig2); hipFree(d_w2L); hipFree(d_w2R); hipFree(d_tmp);
            emagls::pool_stream_give(st);
        };
        try {
            st = emagls::pool_stream_take();
            HIP_CHECK(hipMalloc(&d_in, es_in * nsamp * nch));
            HIP_CHECK(hipMalloc(&d_wL, es_w * len * nch));
            HIP_CHECK(hipMalloc(&d_wR, es_w * len * nch));
            HIP_CHECK(hipMalloc(&d_out, sizeof(double) * nsamp * 2));
            HIP_CHECK(hipMemcpy(d_in, in, es_in * nsamp * nch, hipMemcpyDefault));
            HIP_CHECK(hipMemcpy(d_wL, wL, es_w * len * nch, hipMemcpyDefault));
            HIP_CHECK(hipMemcpy(d_wR, wR, es_w * len * nch, hipMemcpyDefault));
            if (!any_cplx) {
                binaural_decode_real((const double*)d_in, nsamp, (int)nch, (const double*)d_wL, (const double*)d_wR, len, d_out, st);
            } else {
                HIP_CHECK(hipMalloc(&d_sig2, sizeof(double) * 2 * nsamp * nch));
                HIP_CHECK(hipMalloc(&d_w2L, sizeof(double) * 2 * len * nch));
                HIP_CHECK(hipMalloc(&d_w2R, sizeof(double) * 2 * len * nch));
                if (imag_abs_sum) HIP_CHECK(hipMalloc(&d_tmp, sizeof(double) * (2 * nsamp + 2)));
                // (the reference sums the discarded imaginary part AFTER binauralOut(del:end,:), binauralDecode.m:53-62)
                const int64_t cut = (compensate_delay && len / 2 > 0) ? len / 2 - 1 : 0;
                binaural_decode_complex(d_in, in_cplx, nsamp, (int)nch, d_wL, d_wR, w_cplx, len, d_sig2, d_w2L, d_w2R, d_out,
                                        imag_abs_sum, d_tmp, st, std::min(cut, nsamp));
            }
            if (!compensate_delay) {
                HIP_CHECK(hipMemcpy(out, d_out, sizeof(double) * nsamp * 2, hipMemcpyDefault));
            } else {
                // binauralOut(del:end,:), del = len/2 (1-based)   (binauralDecode.m:53-57)
                const int64_t del = len / 2;
                const int64_t skip = del > 0 ? del - 1 : 0;
                const int64_t nout = nsamp - skip;
                if (nout > 0) {
                    HIP_CHECK(hipMemcpy(out, d_out + skip, sizeof(double) * nout, hipMemcpyDefault));
                    HIP_CHECK(hipMemcpy(out + nout, d_out + nsamp + skip, sizeof(double) * nout, hipMemcpyDefault));
                }
            }
        } catch (...) { cleanup(); throw; }
        cleanup();
    });
}

int emagls_binaural_decode_device(const void* d_in, int in_is_complex, int64_t nsamp, int64_t nch, const void* d_wL, const void* d_wR,
                                  int filters_are_complex, int64_t len, double* d_out, double* imag_abs_sum, void* stream) {
    return guarded([&] {
        if (!d_in || !d_wL || !d_wR || !d_out) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (nsamp < 0 || nch < 1 || len < 1) throw Error(EMAGLS_ERR_ARG, "invalid shape");
        if (imag_abs_sum) imag_abs_sum[0] = imag_abs_sum[1] = 0.0;
        if (nsamp == 0) return;
        hipStream_t st = (hipStream_t)stream;
        if (!in_is_complex && !filters_are_complex) {
            binaural_decode_real((const double*)d_in, nsamp, (int)nch, (const double*)d_wL, (const double*)d_wR, len, d_out, st);
            return;
        }
        std::lock_guard<std::mutex> lk(g_decode_scratch.mu);
        g_decode_scratch.ensure((size_t)2 * nsamp * nch, (size_t)2 * len * nch, imag_abs_sum ? (size_t)(2 * nsamp + 2) : 0);
        binaural_decode_complex(d_in, in_is_complex != 0, nsamp, (int)nch, d_wL, d_wR, filters_are_complex != 0, len, g_decode_scratch.sig2,
                                g_decode_scratch.w2L, g_decode_scratch.w2R, d_out, imag_abs_sum, g_decode_scratch.tmp, st, 0);
    });
}

int emagls_get_magls_filters_dc(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi, const double* zen,
                                int order, double fs, int64_t len, int apply_dc, int basis, void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_MAGLS; d.basis = basis; d.order = order; d.fs = fs; d.len = len; d.nsamp = nsamp; d.ndirs = ndirs;
    d.diffuseness = apply_dc != 0;
    return one_shot(d, hL, hR, azi, zen, nullptr, nullptr, nullptr, nullptr, nullptr, wL, wR, nullptr);
}
static int emagls_array_dc(int kind, const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi, const double* zen,
                           double mic_radius, const double* mic_azi, const double* mic_zen, int64_t nmics, int order, double fs, int64_t len,
                           int apply_dc, int basis, void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = kind; d.basis = basis; d.order = order; d.fs = fs; d.len = len; d.nsamp = nsamp; d.ndirs = ndirs;
    d.mic_radius = mic_radius; d.nmics = nmics; d.diffuseness = apply_dc != 0;
    return one_shot(d, hL, hR, azi, zen, mic_azi, mic_zen, nullptr, nullptr, nullptr, wL, wR, nullptr);
}
int emagls_get_emagls_filters_dc(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi, const double* zen,
                                 double mic_radius, const double* mic_azi, const double* mic_zen, int64_t nmics, int order, double fs,
                                 int64_t len, int apply_dc, int basis, void* wL, void* wR) {
    return emagls_array_dc(EMAGLS_KIND_EMAGLS, hL, hR, nsamp, ndirs, azi, zen, mic_radius, mic_azi, mic_zen, nmics, order, fs, len, apply_dc,
                           basis, wL, wR);
}
int emagls_get_emagls2_filters_dc(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi, const double* zen,
                                  double mic_radius, const double* mic_azi, const double* mic_zen, int64_t nmics, int order, double fs,
                                  int64_t len, int apply_dc, int basis, void* wL, void* wR) {
    return emagls_array_dc(EMAGLS_KIND_EMAGLS2, hL, hR, nsamp, ndirs, azi, zen, mic_radius, mic_azi, mic_zen, nmics, order, fs, len, apply_dc,
                           basis, wL, wR);
}
int emagls_get_magls_filters_2d(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi, int order, double fs,
                                int64_t len, int basis, void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_MAGLS_2D; d.basis = basis; d.order = order; d.fs = fs; d.len = len; d.nsamp = nsamp; d.ndirs = ndirs;
    return one_shot(d, hL, hR, azi, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, wL, wR, nullptr);
}

int emagls_simulation_order(int kind, int order, double fs, double mic_radius) {
    switch (kind) {
        case EMAGLS_KIND_EMAGLS: case EMAGLS_KIND_EMA_CH: case EMAGLS_KIND_EMA_SH:
            return std::max(order, (int)std::ceil(fs * kPi * mic_radius / C_SOUND));                   // getSMAIRMatrix.m:95
        case EMAGLS_KIND_EMAGLS2:
            return std::max(SMAIR_DEFAULT_ORDER, (int)std::ceil(fs * kPi * mic_radius / C_SOUND));     // params.order unset -> 4
        default: return order;
    }
}
int emagls_design_out_shape(const emagls_design_desc* desc, int64_t* rows, int64_t* cols, int* is_complex) {
    return guarded([&] {
        if (!desc || !rows || !cols || !is_complex) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        const emagls_design_desc& d = *desc;
        if (d.kind < EMAGLS_KIND_LS || d.kind > EMAGLS_KIND_EMA_SH || d.order < 0) throw Error(EMAGLS_ERR_ARG, "invalid design kind or order");
        switch (d.kind) {
            case EMAGLS_KIND_EMAGLS2: case EMAGLS_KIND_FROM_ATF: *cols = d.nmics; break;                       // one filter per microphone
            case EMAGLS_KIND_MAGLS_2D: case EMAGLS_KIND_EMA_CH: *cols = 2 * (int64_t)d.order + 1; break;       // circular harmonics
            default: *cols = ((int64_t)d.order + 1) * (d.order + 1);
        }
        *rows = d.kind == EMAGLS_KIND_LS ? d.nsamp : d.len;                                                    // lib/getLsFilters.m:33: h * Y_pinv
        *is_complex = d.basis == EMAGLS_BASIS_COMPLEX && d.kind != EMAGLS_KIND_FROM_ATF;
    });
}
int emagls_get_ls_filters_with_basis(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const void* Y_hrir, int order,
                                     int basis, void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_LS; d.basis = basis; d.order = order; d.nsamp = nsamp; d.ndirs = ndirs; d.custom_basis = 1;
    return one_shot(d, hL, hR, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, wL, wR, nullptr, Y_hrir, nullptr);
}
int emagls_get_magls_filters_with_basis(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const void* Y_hrir, int order,
                                        double fs, int64_t len, int basis, void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_MAGLS; d.basis = basis; d.order = order; d.fs = fs; d.len = len; d.nsamp = nsamp; d.ndirs = ndirs; d.custom_basis = 1;
    return one_shot(d, hL, hR, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, wL, wR, nullptr, Y_hrir, nullptr);
}
int emagls_get_emagls_filters_with_basis(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const void* Y_hrir,
                                         double mic_radius, const void* Y_mic, int64_t nmics, int order, double fs, int64_t len, int basis,
                                         void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_EMAGLS; d.basis = basis; d.order = order; d.fs = fs; d.len = len; d.nsamp = nsamp; d.ndirs = ndirs;
    d.mic_radius = mic_radius; d.nmics = nmics; d.custom_basis = 1;
    if (!Y_mic) { g_last_error = "null microphone SH matrix"; return EMAGLS_ERR_ARG; }
    return one_shot(d, hL, hR, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, wL, wR, nullptr, Y_hrir, Y_mic);
}
int emagls_get_emagls2_filters_with_basis(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const void* Y_hrir,
                                          double mic_radius, const void* Y_mic, int64_t nmics, int order, double fs, int64_t len, int basis,
                                          void* wL, void* wR) {
    emagls_design_desc d{};
    d.kind = EMAGLS_KIND_EMAGLS2; d.basis = basis; d.order = order; d.fs = fs; d.len = len; d.nsamp = nsamp; d.ndirs = ndirs;
    d.mic_radius = mic_radius; d.nmics = nmics; d.custom_basis = 1;
    if (!Y_mic) { g_last_error = "null microphone SH matrix"; return EMAGLS_ERR_ARG; }
    return one_shot(d, hL, hR, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, wL, wR, nullptr, Y_hrir, Y_mic);
}

// ---------------------------------------------------------------------------------------------
// HRIR sets on one geometry in ONE call (what a loop over subjects around lib/get*Filters.m does): plans + a geometry-sharing
// batch per chunk of up to 16 sets, kept for the next call of the same shape (emagls_cache_clear releases them).
// ---------------------------------------------------------------------------------------------
namespace {
struct SetsCache {
    emagls_design_desc desc{};
    int device = -1, n = 0;
    std::vector<emagls_plan*> plans;
    emagls_batch* batch = nullptr;
    std::vector<double> grid[4];      // the grids the plans hold (hrir azi / zen, mic azi / zen): unchanged grids are not uploaded again
    void release() {
        for (auto& g : grid) g.clear();
        if (batch) { emagls_batch_destroy(batch); batch = nullptr; }
        for (auto* p : plans) emagls_plan_destroy(p);
        plans.clear();
        n = 0; device = -1;
    }
};
std::mutex g_sets_mu;
SetsCache g_sets[3];   // (two sets of plans for the full chunks, which alternate, and one for the tail chunk)
}  // namespace
void emagls_sets_cache_clear_internal() {
    std::lock_guard<std::mutex> lk(g_sets_mu);
    for (auto& c : g_sets) c.release();
}

int emagls_design_hrir_sets(int kind, const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, int64_t nsets,
                            const double* hrir_azi, const double* hrir_zen, double mic_radius, const double* mic_azi, const double* mic_zen,
                            int64_t nmics, int order, double fs, int64_t len, int basis, void* wL, void* wR) {
    return guarded([&] {
        if (!hL || !hR || !hrir_azi || !wL || !wR || nsets < 1) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        const bool arr = kind == EMAGLS_KIND_EMAGLS || kind == EMAGLS_KIND_EMAGLS2 || kind == EMAGLS_KIND_EMA_CH || kind == EMAGLS_KIND_EMA_SH;
        if (!arr && kind != EMAGLS_KIND_LS && kind != EMAGLS_KIND_MAGLS && kind != EMAGLS_KIND_MAGLS_2D)
            throw Error(EMAGLS_ERR_UNSUPPORTED, "HRIR-set job lists: LS, MagLS, MagLS-2D, eMagLS, eMagLS2, EMAinCH, EMAinSH");
        emagls_design_desc d{};
        d.kind = kind; d.basis = basis; d.order = order; d.fs = fs; d.len = kind == EMAGLS_KIND_LS ? nsamp : len; d.nsamp = nsamp; d.ndirs = ndirs;
        d.mic_radius = arr ? mic_radius : 0.0; d.nmics = arr ? nmics : 0;
        auto req = [](int r) { if (r != EMAGLS_OK) throw Error(r, g_last_error); };
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(g_sets_mu);
        // Full chunks alternate between two plan sets: while one computes, the next chunk's HRIRs are uploaded into the other
        // (5.5 MB per set from pageable memory: as long as the chunk's compute).  The tail chunk has a set of its own.
        struct Pending { SetsCache* c = nullptr; int64_t first = 0; };
        Pending pend[3];
        size_t out_bytes = 0;
        auto collect = [&](Pending& q) {
            if (!q.c) return;
            SetsCache* c = q.c;
            q.c = nullptr;
            const int n = c->n;
            if (n == 1 || !c->batch) {   // (no batch: designs with more than 32 channels run one at a time)
                for (int j = 0; j < n; ++j)
                    req(emagls_plan_get_filters(c->plans[(size_t)j], (char*)wL + (q.first + j) * out_bytes, (char*)wR + (q.first + j) * out_bytes));
            } else {
                std::vector<void*> pl((size_t)n), pr((size_t)n);
                for (int j = 0; j < n; ++j) { pl[(size_t)j] = (char*)wL + (q.first + j) * out_bytes; pr[(size_t)j] = (char*)wR + (q.first + j) * out_bytes; }
                req(emagls_batch_get_filters(c->batch, pl.data(), pr.data()));
            }
        };
        try {
            int64_t k = 0;
            for (int64_t first = 0; first < nsets; ++k) {
                // (chunks of designs with more than 32 channels run plan by plan and hold gigabytes per plan: four at a time)
                const bool wide_kind = ((kind == EMAGLS_KIND_MAGLS_2D || kind == EMAGLS_KIND_EMA_CH) ? 2 * order + 1 : kind == EMAGLS_KIND_EMAGLS2 ? (int)nmics : (order + 1) * (order + 1)) > 32;
                // (eMagLS / eMagLS2 there: one plan per chunk -- two plans alternate and keep their geometry stages, plan_execute -- instead of four)
                const int chunk_max = wide_kind ? ((kind == EMAGLS_KIND_EMAGLS || kind == EMAGLS_KIND_EMAGLS2) ? 1 : 4) : kind == EMAGLS_KIND_EMA_SH ? 4 : SWEEP_MULTI_MAX;
                const int n = (int)std::min<int64_t>(chunk_max, nsets - first);
                const int slot = n == chunk_max ? (int)(k % 2) : 2;
                SetsCache* c = &g_sets[slot];
                collect(pend[slot]);      // (the chunk this plan set computed two chunks ago)
                if (!(c->n == n && c->device == dev && same_desc(c->desc, d))) {
                    c->release();
                    try {
                        for (int j = 0; j < n; ++j) {
                            emagls_plan* p = nullptr;
                            req(emagls_plan_create(&d, &p));
                            p->geo_keep = true;   // (one geometry for every set by this entry point's contract: plans of the 33..64-channel path keep their geometry stages)
                            c->plans.push_back(p);
                        }
                        // designs with more than 32 channels (LS / MagLS orders 5..7, arrays of 33..64 channels) do not enter
                        // batches (emagls_batch_create): their chunk runs plan by plan, same filters as nsets single calls
                        // (EMAinSH -- lib/getEMagLsFiltersEMAinSH.m:32 -- has no lane batches either: plan by plan)
                        if (n > 1 && !c->plans[0]->wide && kind != EMAGLS_KIND_EMA_SH) {
                            g_batch_max_override = SWEEP_MULTI_MAX;
                            const int r = emagls_batch_create(c->plans.data(), n, &c->batch);
                            g_batch_max_override = 0;
                            req(r);
                            req(emagls_batch_set_geometry_sharing(c->batch, 1));
                        }
                    } catch (...) { g_batch_max_override = 0; c->release(); throw; }
                    c->desc = d; c->device = dev; c->n = n;
                }
                auto same = [](const std::vector<double>& have, const double* now, size_t cnt) {
                    return now ? (have.size() == cnt && std::memcmp(have.data(), now, cnt * sizeof(double)) == 0) : have.empty();
                };
                const bool hgrid_same = same(c->grid[0], hrir_azi, (size_t)ndirs) && same(c->grid[1], hrir_zen, (size_t)ndirs) && !c->grid[0].empty();
                const bool mgrid_same = !arr || (same(c->grid[2], mic_azi, (size_t)nmics) && same(c->grid[3], mic_zen, (size_t)nmics) && !c->grid[2].empty());
                for (int j = 0; j < n; ++j) {
                    emagls_plan* p = c->plans[(size_t)j];
                    if (!hgrid_same) req(emagls_plan_set_hrir_grid(p, hrir_azi, hrir_zen));
                    if (arr && !mgrid_same) req(emagls_plan_set_mic_grid(p, mic_azi, mic_zen));
                    req(emagls_plan_set_hrirs(p, hL + (first + j) * nsamp * ndirs, hR + (first + j) * nsamp * ndirs));
                }
                auto keep = [](std::vector<double>& dst, const double* src, size_t cnt) { if (src) dst.assign(src, src + cnt); else dst.clear(); };
                keep(c->grid[0], hrir_azi, (size_t)ndirs); keep(c->grid[1], hrir_zen, (size_t)ndirs);
                if (arr) { keep(c->grid[2], mic_azi, (size_t)nmics); keep(c->grid[3], mic_zen, (size_t)nmics); }
                emagls_plan_info info;
                req(emagls_plan_get_info(c->plans[0], &info));
                out_bytes = (info.out_is_complex ? sizeof(cplx) : sizeof(double)) * (size_t)info.out_rows * info.out_cols;
                if (n > 1 && c->batch) req(emagls_batch_execute(c->batch));   // (asynchronous)
                else for (int j = 0; j < n; ++j) req(emagls_plan_execute(c->plans[(size_t)j]));
                pend[slot].c = c;
                pend[slot].first = first;
                first += n;
            }
            for (auto& q : pend) collect(q);
        } catch (...) {   // (a failed call leaves the plans in an unknown state)
            for (auto& c : g_sets) c.release();
            throw;
        }
    });
}

// The HRTF subjects of ONE ATF set in one call (BASELINE config 5; lib/getEMagLsFiltersFromAtf.m:1 in a loop over subjects).  The
// ATF set is uploaded once (plan 0) and handed to the other plans device to device; the batch then finds equal ATF sides and
// computes that side once (batch_atf_decide_sharing).
namespace {
std::mutex g_atfsets_mu;
SetsCache g_atfsets[2];
}  // namespace
void emagls_atfsets_cache_clear_internal() {
    std::lock_guard<std::mutex> lk(g_atfsets_mu);
    for (auto& c : g_atfsets) c.release();
}

// ---------------------------------------------------------------------------------------------
// Job lists (independent designs are the unit of parallelism -- the loop over array radii / HRIR sets / subjects that
// a user of the reference writes around one of its functions, testEMagLs.m:75-95).  emagls_jobs_run takes the list, cuts it into
// chunks of consecutive jobs of one shape, and keeps up to `in_flight` chunks between input upload and result collection: every
// chunk in flight has a worker thread of the library (uploads, the batch's graph launches, the wait for its filters), so that one
// chunk's inputs travel and another's results are collected while the GPU works on the others.  The plans and lane batches of a
// chunk shape stay resident between calls (released by emagls_cache_clear) whenever the chunk's designs can be re-used as they are
// (same descriptors); array radii that change from chunk to chunk get plans of their own and release them.
// ---------------------------------------------------------------------------------------------
namespace {
struct JobSlot {
    std::string key;                  // the descriptors of the slot's designs, byte for byte
    std::string shape;                // what makes designs share a lane batch (job_shape), job by job: a slot of another key but this shape hands its memory on
    int device = -1;
    std::vector<emagls_plan*> plans;
    emagls_batch* batch = nullptr;
    std::vector<std::vector<double>> grids;   // per plan: hrir azi | zen | mic azi | zen as last uploaded (unchanged grids are not uploaded again)
    uint64_t last_use = 0, last_call = 0;     // (last_call: the emagls_jobs_run call that used the slot last)
    int runs = 0;                             // executes so far (the first two are the eager run and the graph capture)
    hipStream_t stream = nullptr;             // the plans' common stream (uploads, and the executes of designs that run plan by plan)
    ~JobSlot() {
        if (batch) emagls_batch_destroy(batch);
        for (auto* p : plans) emagls_plan_destroy(p);
        if (stream) emagls::pool_stream_give(stream);   // (synchronised there)
    }
};
std::mutex g_jobs_mu;
std::vector<std::unique_ptr<JobSlot>> g_jobs_free;   // resident slots nobody uses at the moment
std::atomic<int> g_jobs_prof{0};                     // emagls_jobs_set_profiling: the chunks' batches time their sweep launches
// A resident chunk's second run (the hipGraph capture of the stages around the sweep) has the job lists' share of the device to itself:
// captures next to other threads' uploads or launches ended in hipErrorStreamCaptureInvalidated ("operation failed due to a previous
// error during capture").  Every other run shares the lock.
std::shared_timed_mutex g_jobs_warm_mu;
uint64_t g_jobs_tick = 0;
std::atomic<size_t> g_jobs_resident_max{8 * REG_SWEEP_MAX};   // designs kept resident between calls (0.19 GB each at config 3); at least one call's chunks in flight

void check_rc(int rc) { if (rc != EMAGLS_OK) throw Error(rc, g_last_error); }
bool is_device_pointer(const void* p) {
    hipPointerAttribute_t at{};
    if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // (plain host memory: not registered)
    return at.type == hipMemoryTypeDevice;
}
// what makes two designs share a lane batch: everything but the array radius inside one (padded) simulation-order class
void job_shape(const emagls_design_desc& d, std::string& out) {
    emagls_design_desc k = d;
    if (array_kind(d.kind) && d.sim_order_pad > 0) {
        const int own = std::max(d.kind == EMAGLS_KIND_EMAGLS2 ? SMAIR_DEFAULT_ORDER : d.order, (int)std::ceil(d.fs * kPi * d.mic_radius / C_SOUND));
        if (own <= d.sim_order_pad) k.mic_radius = 0.0;   // (laid out for sim_order_pad whatever the radius)
    }
    out.assign(reinterpret_cast<const char*>(&k), sizeof k);
}
// will the slot's next execute capture hipGraphs?  (the batch forms and plan_execute capture on the run after an eager one)
bool slot_will_capture(const JobSlot& s) {
    if (s.batch) {
        const emagls_batch& b = *s.batch;
        return b.use_graph && b.eager_runs >= 1 && !b.graph_exec && !(b.plans.size() && b.plans[0]->pre_exec);
    }
    for (const emagls_plan* p : s.plans)
        if (p->use_graph && p->prof_level == 0 && p->eager_runs >= 1 && !p->pre_exec && !p->graph_exec) return true;
    return false;
}
void jobs_run_chunk(const emagls_job* jobs, int n, int device, int flags, bool solo, uint64_t call) {
    DeviceGuard dg(device);
    static const bool trace = getenv("EMAGLS_JOBS_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (trace) fprintf(stderr, "emagls jobs: chunk of %d, %s at %.3f ms\n", n, what,
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    // (a list of ONE chunk has the device to itself: its batch runs the stages before the sweep as one lane group forked onto three
    // streams -- the independent branches of the pipeline side by side: 2160 against 2070 sets/s for a list of 20 config-3 designs --,
    // chunks that share the device with others as two single-stream lane groups: 3300 against 3140 sets/s at 512 designs)
    std::string key(solo ? "S" : "P"), shape(solo ? "S" : "P"), one;
    for (int j = 0; j < n; ++j) {
        key.append(reinterpret_cast<const char*>(&jobs[j].desc), sizeof(emagls_design_desc));
        job_shape(jobs[j].desc, one);
        shape.append(one);
    }
    std::unique_ptr<JobSlot> slot, recycled;
    {
        std::lock_guard<std::mutex> lk(g_jobs_mu);
        for (size_t i = 0; i < g_jobs_free.size(); ++i)
            if (g_jobs_free[i]->device == device && g_jobs_free[i]->key == key) { slot = std::move(g_jobs_free[i]); g_jobs_free.erase(g_jobs_free.begin() + i); break; }
        // no resident slot of these designs, but an idle one of the same SHAPE (other array radii of the same padded classes: the next
        // list of a radius study): its memory serves the new slot -- slabs and arena go back to the block pool and come out again at
        // exactly the sizes asked for.  Fresh device memory is what a new slot must not need: hipMalloc of a 4 GB arena took 0.3 ms on
        // some boxes and 0.5 s on others (the driver clears VRAM it hands out for the first time), 1 s per call of 28 new radii.
        if (!slot)
            for (size_t i = 0; i < g_jobs_free.size(); ++i)
                if (g_jobs_free[i]->device == device && g_jobs_free[i]->shape == shape && g_jobs_free[i]->last_call != call) {   // (not a chunk of THIS list: a repeat of the list finds all its chunks resident)
                    recycled = std::move(g_jobs_free[i]); g_jobs_free.erase(g_jobs_free.begin() + i); break;
                }
    }
    recycled.reset();   // (outside the lock: the destructor waits for the slot's streams)
    // a chunk shares the device with the other chunks in flight, except on its slot's SECOND run: that one captures the hipGraphs of the
    // stages around the sweep, and a capture next to another thread's uploads or launches is invalidated (hipErrorStreamCaptureInvalidated)
    // (decided from the objects' own state, not from the slot's run count: a batch whose graphs were dropped by a recovery --
    // drop_batch_graphs after a flagged bin or a lanes rebuild -- captures again on a later run)
    const bool capturing = slot && slot_will_capture(*slot);
    std::shared_lock<std::shared_timed_mutex> shared(g_jobs_warm_mu, std::defer_lock);
    std::unique_lock<std::shared_timed_mutex> alone(g_jobs_warm_mu, std::defer_lock);
    // (the exclusive lock only around the execute that captures: the uploads before it and the wait for the filters after it run next to
    // the other chunks -- a first call with host arrays spent 40 ... 140 ms per chunk uploading under the exclusive lock)
    shared.lock();
    if (!slot) {
        slot.reset(new JobSlot);
        slot->key = key; slot->shape = shape; slot->device = device;
        slot->grids.resize((size_t)n);
        static const bool share_stream = [] { const char* e = getenv("EMAGLS_JOBS_SHARED_STREAM"); return !(e && e[0] == '0'); }();
        if (share_stream) slot->stream = emagls::pool_stream_take();
        struct SharedStream { SharedStream(hipStream_t st) { g_plan_stream_shared = st; } ~SharedStream() { g_plan_stream_shared = nullptr; } } shared_stream(slot->stream);
        for (int j = 0; j < n; ++j) {
            emagls_plan* p = nullptr;
            check_rc(emagls_plan_create(&jobs[j].desc, &p));
            slot->plans.push_back(p);
            if (j == 0) lap("first plan created");
        }
        lap("plans created");
    }
    for (int j = 0; j < n; ++j) {
        const emagls_job& jb = jobs[j];
        emagls_plan* p = slot->plans[(size_t)j];
        const emagls_design_desc& d = jb.desc;
        // grids (host arrays): uploaded when they differ from what the plan holds
        std::vector<double> g;
        auto app = [&](const double* a, int64_t m) { if (a) g.insert(g.end(), a, a + m); else g.push_back(-1e300); };
        const bool has_mics = array_kind(d.kind);
        app(jb.hrir_azi, d.ndirs); app(jb.hrir_zen, d.ndirs);
        if (has_mics) { app(jb.mic_azi, d.nmics); app(jb.mic_zen, d.nmics); }
        if (g != slot->grids[(size_t)j]) {
            if (!jb.hrir_azi) throw Error(EMAGLS_ERR_ARG, "job without an HRIR grid");
            check_rc(emagls_plan_set_hrir_grid(p, jb.hrir_azi, jb.hrir_zen));
            if (has_mics) {
                if (!jb.mic_azi) throw Error(EMAGLS_ERR_ARG, "array design without a microphone grid");
                check_rc(emagls_plan_set_mic_grid(p, jb.mic_azi, jb.mic_zen));
            }
            slot->grids[(size_t)j] = std::move(g);
        }
        if (d.kind == EMAGLS_KIND_FROM_ATF) {
            if (!jb.atf || !jb.atf_azi || !jb.atf_zen) throw Error(EMAGLS_ERR_ARG, "FromAtf job without its ATF set");
            check_rc(emagls_plan_set_atfs(p, jb.atf, jb.atf_azi, jb.atf_zen));
        }
        if (!jb.hL || !jb.hR || !jb.wL || !jb.wR) throw Error(EMAGLS_ERR_ARG, "job without HRIRs or without room for its filters");
        if (!slot->batch && n > 1 && !slot->runs) check_rc(emagls_plan_set_hrirs(p, jb.hL, jb.hR));   // (before the batch exists: plan by plan)
    }
    lap("inputs set");
    if (n > 1 && !slot->batch && !slot->runs) {
        g_batch_max_override = REG_SWEEP_MAX;
        const int rc = emagls_batch_create(slot->plans.data(), n, &slot->batch);
        g_batch_max_override = 0;
        if (rc != EMAGLS_OK && rc != EMAGLS_ERR_UNSUPPORTED) check_rc(rc);   // (unsupported as a batch -- e.g. more than 32 channels: plan by plan)
        if (rc != EMAGLS_OK) slot->batch = nullptr;
        if (slot->batch && solo && slot->batch->lanes) {
            static const int fork = [] { const char* e = getenv("EMAGLS_JOBS_FORK"); return e ? std::max(1, std::min(4, atoi(e))) : 3; }();
            if (fork >= 2) {
                if (slot->batch->groups != 1) { HIP_CHECK(hipStreamSynchronize(slot->batch->stream)); drop_batch_graphs(*slot->batch); slot->batch->groups = 1; }
                check_rc(emagls_batch_set_streams(slot->batch, fork));
            }
        }
        if (slot->batch) {
            slot->batch->alone = solo;   // (chunks in flight next to each other keep the orthonormal route before their sweeps: batch_defers_hh)
            if (!solo && slot->batch->graph_exec) { HIP_CHECK(hipStreamSynchronize(slot->batch->stream)); drop_batch_graphs(*slot->batch); }
        }
        lap("batch created");
    } else if (slot->batch) {
        // the HRIRs of every design on the batch's stream, ordered before its execute: no host synchronisation per plan
        // (jobs that name the SAME HRIR arrays -- one HRIR set for every radius of an array sweep -- are served device to device from
        // the first plan that received them: 5.5 MB over PCIe instead of 5.5 MB per design from pageable memory)
        // inputs that already lie in device memory: ONE gather launch for the whole chunk (hipMemcpyAsync per buffer otherwise)
        bool gathered = false;
        if (2 * n <= 64) {
            bool all_dev = true;
            for (int j = 0; j < n && all_dev; ++j) all_dev = is_device_pointer(jobs[j].hL) && is_device_pointer(jobs[j].hR);
            if (all_dev) {
                LanePtrs src{}, dst{};
                const emagls_plan& q0 = *slot->plans[0];
                const size_t bytes = sizeof(double) * (size_t)q0.d.nsamp * (size_t)q0.d.ndirs;
                for (int j = 0; j < n; ++j) {
                    emagls_plan& q = *slot->plans[(size_t)j];
                    src.p[2 * j] = const_cast<double*>(jobs[j].hL); src.p[2 * j + 1] = const_cast<double*>(jobs[j].hR);
                    dst.p[2 * j] = q.get("hL"); dst.p[2 * j + 1] = q.get("hR");
                    q.have_hrirs = true;
                }
                launch_gather_buffers(src, dst, 2 * n, bytes, slot->batch->stream);
                gathered = true;
            }
        }
        for (int j = 0; j < n && !gathered; ++j) {
            emagls_plan& q = *slot->plans[(size_t)j];
            const size_t bytes = sizeof(double) * (size_t)q.d.nsamp * (size_t)q.d.ndirs;
            int src = -1;
            for (int i = 0; i < j && src < 0; ++i)
                if (jobs[i].hL == jobs[j].hL && jobs[i].hR == jobs[j].hR && slot->plans[(size_t)i]->d.nsamp == q.d.nsamp && slot->plans[(size_t)i]->d.ndirs == q.d.ndirs) src = i;
            const void* sl = src >= 0 ? slot->plans[(size_t)src]->get("hL") : jobs[j].hL;
            const void* sr = src >= 0 ? slot->plans[(size_t)src]->get("hR") : jobs[j].hR;
            HIP_CHECK(hipMemcpyAsync(q.get("hL"), sl, bytes, hipMemcpyDefault, slot->batch->stream));
            HIP_CHECK(hipMemcpyAsync(q.get("hR"), sr, bytes, hipMemcpyDefault, slot->batch->stream));
            q.have_hrirs = true;
        }
    } else {
        for (int j = 0; j < n; ++j) check_rc(emagls_plan_set_hrirs(slot->plans[(size_t)j], jobs[j].hL, jobs[j].hR));
    }
    if (slot->batch && (flags & EMAGLS_JOBS_SHARE_GEOMETRY)) {
        // HRIR sets on one geometry: the geometry stages once per chunk (the library compares the grids on the device, and a chunk whose
        // designs do not agree runs them as independent designs -- emagls_batch_set_geometry_sharing)
        const int kind = jobs[0].desc.kind;
        if (kind != EMAGLS_KIND_FROM_ATF && kind != EMAGLS_KIND_EMA_SH) check_rc(emagls_batch_set_geometry_sharing(slot->batch, 1));
    }
    // (designs of the 33..64-channel path run plan by plan: with the flag a plan keeps its geometry stages from its last clean run while
    // its own grids stay the same -- plan_execute)
    if (!slot->batch) for (auto* q : slot->plans) q->geo_keep = (flags & EMAGLS_JOBS_SHARE_GEOMETRY) != 0;
    {
        if (slot->batch) {
            std::vector<void*> wl((size_t)n), wr((size_t)n);
            for (int j = 0; j < n; ++j) { wl[(size_t)j] = jobs[j].wL; wr[(size_t)j] = jobs[j].wR; }
            if (slot->batch->prof_level != g_jobs_prof.load()) check_rc(emagls_batch_set_profiling(slot->batch, g_jobs_prof.load()));
            if (capturing) { shared.unlock(); alone.lock(); }
            check_rc(emagls_batch_execute(slot->batch));
            if (capturing) { alone.unlock(); shared.lock(); }
            check_rc(emagls_batch_get_filters(slot->batch, wl.data(), wr.data()));
        } else {
            if (capturing) { shared.unlock(); alone.lock(); }
            for (int j = 0; j < n; ++j) check_rc(emagls_plan_execute(slot->plans[(size_t)j]));
            if (capturing) { alone.unlock(); shared.lock(); }
            for (int j = 0; j < n; ++j) check_rc(emagls_plan_get_filters(slot->plans[(size_t)j], jobs[j].wL, jobs[j].wR));
        }
        ++slot->runs;
    }
    lap("executed and collected");
    // keep the slot when its designs can serve another chunk as they are
    std::lock_guard<std::mutex> lk(g_jobs_mu);
    slot->last_use = ++g_jobs_tick;
    slot->last_call = call;
    g_jobs_free.push_back(std::move(slot));
    size_t resident = 0;
    for (auto& f : g_jobs_free) resident += f->plans.size();
    while (resident > g_jobs_resident_max.load() && g_jobs_free.size() > 1) {   // least recently used first
        size_t old = 0;
        for (size_t i = 1; i < g_jobs_free.size(); ++i) if (g_jobs_free[i]->last_use < g_jobs_free[old]->last_use) old = i;
        resident -= g_jobs_free[old]->plans.size();
        g_jobs_free.erase(g_jobs_free.begin() + old);
    }
}
}  // namespace
void emagls_jobs_cache_clear_internal() {
    std::lock_guard<std::mutex> lk(g_jobs_mu);
    g_jobs_free.clear();
}

int emagls_jobs_set_profiling(int level) {
    return guarded([&] { g_jobs_prof.store(level > 0 ? 1 : 0); });
}
int emagls_jobs_sweep_times(double* ms, int* designs, int capacity, int* count) {
    return guarded([&] {
        if (!count || capacity < 0 || (capacity > 0 && (!ms || !designs))) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        std::lock_guard<std::mutex> lk(g_jobs_mu);
        int n = 0;
        for (auto& f : g_jobs_free) {
            if (!f->batch || f->batch->prof_level < 1 || !f->batch->sweep_ev[0] || f->batch->eager_runs < 1) continue;
            if (n < capacity) {
                DeviceGuard dg(f->device);
                float t = 0.f;
                if (hipEventElapsedTime(&t, f->batch->sweep_ev[0], f->batch->sweep_ev[1]) != hipSuccess) { (void)hipGetLastError(); continue; }
                ms[n] = t; designs[n] = (int)f->plans.size();
            }
            ++n;
        }
        *count = n;
    });
}

int emagls_jobs_run(const emagls_job* jobs, int64_t njobs, int batch_size, int in_flight, int flags) {
    return guarded([&] {
        if (!jobs || njobs < 0) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        if (njobs == 0) return;
        if (batch_size <= 0) batch_size = REG_SWEEP_MAX;
        if (in_flight <= 0) in_flight = 4;
        batch_size = std::min(batch_size, REG_SWEEP_MAX);
        g_jobs_resident_max.store(std::max((size_t)8 * REG_SWEEP_MAX, (size_t)2 * batch_size * (size_t)in_flight));   // (EMAGLS_JOBS_RESIDENT overrides)
        if (const char* e = getenv("EMAGLS_JOBS_RESIDENT")) { const long v = atol(e); if (v > 0) g_jobs_resident_max.store((size_t)v); }
        int device = 0;
        HIP_CHECK(hipGetDevice(&device));
        // chunks: consecutive jobs of one shape; more than 16 designs per chunk only where the register-resident sweep takes them
        // (array designs on the built-in basis; decided when the batch is created: a refused batch of 17 ... 32 is an error the caller
        // avoids by asking for batches of 16)
        std::vector<std::pair<int64_t, int>> chunks;
        for (int64_t first = 0; first < njobs;) {
            std::string shape, other;
            job_shape(jobs[first].desc, shape);
            int cap = array_kind(jobs[first].desc.kind) && !jobs[first].desc.custom_basis && !jobs[first].desc.diffuseness ? batch_size
                                                                                                                            : std::min(batch_size, SWEEP_MULTI_MAX);
            // HRIR sets on one geometry on the 33..64-channel path (plan by plan, gigabytes per plan): one design per chunk, so that the sets
            // pass through `in_flight` plans which keep their geometry stages (plan_execute) instead of one plan per set
            {
                const emagls_design_desc& d0 = jobs[first].desc;
                const int64_t ch = d0.kind == EMAGLS_KIND_EMAGLS2 ? d0.nmics : (int64_t)(d0.order + 1) * (d0.order + 1);
                if ((flags & EMAGLS_JOBS_SHARE_GEOMETRY) && (d0.kind == EMAGLS_KIND_EMAGLS || d0.kind == EMAGLS_KIND_EMAGLS2) && ch > 32) cap = 1;
            }
            int n = 1;
            while (first + n < njobs && n < cap && (job_shape(jobs[first + n].desc, other), other == shape)) ++n;
            chunks.emplace_back(first, n);
            first += n;
        }
        static std::atomic<uint64_t> g_jobs_call{0};
        const uint64_t call = ++g_jobs_call;
        // workers: each takes the next chunk until none is left; the first error stops the hand-out and is reported
        std::atomic<size_t> next{0};
        std::mutex err_mu;
        int err_code = EMAGLS_OK;
        std::string err_msg;
        auto work = [&] {
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= chunks.size()) return;
                {
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (err_code != EMAGLS_OK) return;
                }
                try {
                    try {
                        jobs_run_chunk(jobs + chunks[c].first, chunks[c].second, device, flags, chunks.size() == 1, call);
                    } catch (const Error& e) {
                        // a chunk of 17 ... 32 designs whose sweep stopped being the register-resident form (a recovery moved it to
                        // the slab or launch-per-bin forms, which hold 16 designs): the same designs as two chunks of at most 16
                        if (chunks[c].second <= SWEEP_MULTI_MAX || e.code != EMAGLS_ERR_UNSUPPORTED || !strstr(e.what(), "more than 16 designs")) throw;
                        const int h = (chunks[c].second + 1) / 2;
                        jobs_run_chunk(jobs + chunks[c].first, h, device, flags, false, call);
                        jobs_run_chunk(jobs + chunks[c].first + h, chunks[c].second - h, device, flags, false, call);
                    }
                } catch (const Error& e) {
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (err_code == EMAGLS_OK) { err_code = e.code; err_msg = e.what(); }
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> lk(err_mu);
                    if (err_code == EMAGLS_OK) { err_code = EMAGLS_ERR_HIP; err_msg = e.what(); }
                }
            }
        };
        const int nthreads = (int)std::min<size_t>((size_t)in_flight, chunks.size());
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
        if (err_code != EMAGLS_OK) throw Error(err_code, err_msg);
    });
}
// ---------------------------------------------------------------------------------------------
// Job lists over several GPUs at the C boundary: the split of emagls_amd/batch.py restated in C, and a runner that
// drives the devices of ONE process from a thread each -- what a MEX caller (one MATLAB process) or any C host has without
// torch.distributed.  One process per GPU with an RCCL gather is the other form (emagls_amd/batch.py; INTEGRATION.md shows both).
// ---------------------------------------------------------------------------------------------
namespace {
// run time model of one lane batch of n designs laid out for sim_order (emagls_amd/batch.py: batch_cost, measured in round 4)
double shard_batch_cost(int n, int sim_order) {
    const double S = (double)(sim_order + 1) * (sim_order + 1), g = n / 8.0;
    if (sim_order < 19) return 5.6 + 7.2 * g;
    return (6.4 + 2.4 * g) + (1.67 + 1.93 * g) * 1e-3 * S;
}
// emagls_amd/batch.py: padded_lane_batches(sim_orders, max_batch, balance=True) -- consecutive chunks of the jobs sorted by simulation
// order, cut at equal COST; returns (first, size) into `order`
std::vector<std::pair<int, int>> shard_padded_batches(const std::vector<int>& so_sorted, int max_batch) {
    const int n = (int)so_sorted.size();
    const int nb = (n + max_batch - 1) / max_batch;
    std::vector<std::pair<int, int>> out;
    if (nb <= 1) { out.emplace_back(0, n); return out; }
    auto cut = [&](double T, std::vector<std::pair<int, int>>& chunks) {
        chunks.clear();
        int pos = 0;
        while (pos < n) {
            if (shard_batch_cost(1, so_sorted[pos]) > T) return false;
            int size = 1;
            while (pos + size < n && size < 32 && shard_batch_cost(size + 1, so_sorted[pos + size]) <= T) ++size;
            chunks.emplace_back(pos, size);
            pos += size;
        }
        return true;
    };
    double lo = 0.0, hi = shard_batch_cost(32, so_sorted.back()) + 1.0;
    std::vector<std::pair<int, int>> c;
    for (int it = 0; it < 40; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (cut(mid, c) && (int)c.size() <= nb) hi = mid; else lo = mid;
    }
    cut(hi, out);
    return out;
}
struct ShardUnit { std::vector<int> jobs; int pad = 0; double cost = 1.0; };
// the units of a list (lane batches of array-radius families, single jobs otherwise) in list order of their first job
void shard_units(const emagls_job* jobs, int64_t njobs, int max_batch, std::vector<ShardUnit>& units) {
    // families: everything but the array radius (and the padding) equal
    std::map<std::string, std::vector<int>> fam;
    std::vector<std::string> fam_order;
    for (int64_t j = 0; j < njobs; ++j) {
        emagls_design_desc k = jobs[j].desc;
        const bool radius_family = (k.kind == EMAGLS_KIND_EMAGLS || k.kind == EMAGLS_KIND_EMAGLS2 || k.kind == EMAGLS_KIND_EMA_CH) && !k.custom_basis && k.nmics <= 32;
        if (radius_family) { k.mic_radius = 0.0; k.sim_order_pad = 0; }
        std::string key(reinterpret_cast<const char*>(&k), sizeof k);
        key.push_back(radius_family ? 'R' : 'E');
        if (!fam.count(key)) fam_order.push_back(key);
        fam[key].push_back((int)j);
    }
    for (const std::string& key : fam_order) {
        const std::vector<int>& idx = fam[key];
        if (key.back() == 'R' && idx.size() > 1) {
            std::vector<int> order(idx.size());
            for (size_t i = 0; i < idx.size(); ++i) order[i] = (int)i;
            auto so_of = [&](int i) { const emagls_design_desc& d = jobs[idx[(size_t)i]].desc; return emagls_simulation_order(d.kind, d.order, d.fs, d.mic_radius); };
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return so_of(a) < so_of(b); });
            std::vector<int> so_sorted(order.size());
            for (size_t i = 0; i < order.size(); ++i) so_sorted[i] = so_of(order[i]);
            if (so_sorted.front() == so_sorted.back()) {   // one simulation-order class: equal jobs, a unit each (the library cuts a rank's share into chunks)
                for (int j : idx) { ShardUnit u; u.jobs.push_back(j); u.cost = 1.0; units.push_back(std::move(u)); }
                continue;
            }
            for (auto& fs : shard_padded_batches(so_sorted, max_batch)) {
                ShardUnit u;
                for (int i = fs.first; i < fs.first + fs.second; ++i) { u.jobs.push_back(idx[(size_t)order[(size_t)i]]); u.pad = std::max(u.pad, so_sorted[(size_t)i]); }
                u.cost = shard_batch_cost((int)u.jobs.size(), u.pad);
                units.push_back(std::move(u));
            }
        } else {
            for (int j : idx) { ShardUnit u; u.jobs.push_back(j); u.cost = 1.0; units.push_back(std::move(u)); }
        }
    }
}
}  // namespace

int emagls_jobs_shard(const emagls_job* jobs, int64_t njobs, int world, int max_batch, int* rank_of_job, int* order_in_rank, int* sim_order_pad) {
    return guarded([&] {
        if (!jobs || njobs < 0 || world < 1 || !rank_of_job) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        if (max_batch <= 0) max_batch = 16;
        if (max_batch > 32) throw Error(EMAGLS_ERR_ARG, "a batch holds 1..32 designs");
        std::vector<ShardUnit> units;
        shard_units(jobs, njobs, max_batch, units);
        // whole units to ranks by longest processing time (ties: the earlier unit, the lower rank), cheapest first inside a rank
        std::vector<int> by_cost(units.size());
        for (size_t i = 0; i < units.size(); ++i) by_cost[i] = (int)i;
        std::stable_sort(by_cost.begin(), by_cost.end(), [&](int a, int b) { return units[(size_t)a].cost > units[(size_t)b].cost; });
        std::vector<double> load((size_t)world, 0.0);
        std::vector<std::vector<int>> mine((size_t)world);
        for (int u : by_cost) {
            int r = 0;
            for (int q = 1; q < world; ++q) if (load[(size_t)q] < load[(size_t)r]) r = q;
            mine[(size_t)r].push_back(u);
            load[(size_t)r] += units[(size_t)u].cost;
        }
        for (int r = 0; r < world; ++r) {
            std::stable_sort(mine[(size_t)r].begin(), mine[(size_t)r].end(), [&](int a, int b) {
                return units[(size_t)a].cost < units[(size_t)b].cost || (units[(size_t)a].cost == units[(size_t)b].cost && a < b); });
            int pos = 0;
            for (int u : mine[(size_t)r])
                for (int j : units[(size_t)u].jobs) {
                    rank_of_job[j] = r;
                    if (order_in_rank) order_in_rank[j] = pos;
                    if (sim_order_pad) sim_order_pad[j] = units[(size_t)u].pad;
                    ++pos;
                }
        }
    });
}

int emagls_jobs_run_devices(const emagls_job* jobs, int64_t njobs, const int* devices, int ndevices, int batch_size, int in_flight, int flags) {
    return guarded([&] {
        if (!jobs || njobs < 0 || !devices || ndevices < 1) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        int avail = 0;
        HIP_CHECK(hipGetDeviceCount(&avail));
        for (int i = 0; i < ndevices; ++i) if (devices[i] < 0 || devices[i] >= avail) throw Error(EMAGLS_ERR_ARG, "no such device");
        if (njobs == 0) return;
        std::vector<int> rank((size_t)njobs), pos((size_t)njobs), pad((size_t)njobs);
        check_rc(emagls_jobs_shard(jobs, njobs, ndevices, std::min(batch_size > 0 ? batch_size : 16, 16), rank.data(), pos.data(), pad.data()));
        // every device's share as a list of its own (jobs of one lane batch adjacent, laid out for the batch's simulation order)
        std::vector<std::vector<emagls_job>> share((size_t)ndevices);
        for (int r = 0; r < ndevices; ++r) {
            int64_t cnt = 0;
            for (int64_t j = 0; j < njobs; ++j) cnt += rank[(size_t)j] == r;
            share[(size_t)r].resize((size_t)cnt);
        }
        for (int64_t j = 0; j < njobs; ++j) {
            emagls_job jb = jobs[j];
            if (pad[(size_t)j] > 0 && jb.desc.sim_order_pad == 0) jb.desc.sim_order_pad = pad[(size_t)j];
            share[(size_t)rank[(size_t)j]][(size_t)pos[(size_t)j]] = jb;
        }
        std::vector<int> rc((size_t)ndevices, EMAGLS_OK);
        std::vector<std::string> msg((size_t)ndevices);
        auto work = [&](int r) {
            if (share[(size_t)r].empty()) return;
            if (hipSetDevice(devices[r]) != hipSuccess) { rc[(size_t)r] = EMAGLS_ERR_HIP; msg[(size_t)r] = "hipSetDevice failed"; return; }
            rc[(size_t)r] = emagls_jobs_run(share[(size_t)r].data(), (int64_t)share[(size_t)r].size(), batch_size > 0 ? batch_size : 32, in_flight, flags);
            if (rc[(size_t)r] != EMAGLS_OK) msg[(size_t)r] = g_last_error;   // (thread-local: this thread's)
        };
        std::vector<std::thread> th;
        for (int r = 1; r < ndevices; ++r) th.emplace_back(work, r);
        int keep = 0;
        HIP_CHECK(hipGetDevice(&keep));
        work(0);
        for (auto& t : th) t.join();
        HIP_CHECK(hipSetDevice(keep));
        for (int r = 0; r < ndevices; ++r)
            if (rc[(size_t)r] != EMAGLS_OK) throw Error(rc[(size_t)r], "device " + std::to_string(devices[r]) + ": " + msg[(size_t)r]);
    });
}

int emagls_from_atf_hrir_sets(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, int64_t nsets, const double* hrir_azi,
                              const double* hrir_zen, const double* atf_irs, int64_t atf_taps, int64_t nmics, int64_t natf, const double* atf_azi,
                              const double* atf_zen, double fs, int64_t filter_len, double f_trans, double* wL, double* wR, double* mean_dev) {
    return guarded([&] {
        if (!hL || !hR || !hrir_azi || !hrir_zen || !atf_irs || !atf_azi || !atf_zen || !wL || !wR || nsets < 1) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        emagls_design_desc d{};
        d.kind = EMAGLS_KIND_FROM_ATF; d.basis = EMAGLS_BASIS_REAL; d.fs = fs; d.len = filter_len; d.nsamp = nsamp; d.ndirs = ndirs;
        d.nmics = nmics; d.f_trans = f_trans; d.atf_taps = atf_taps; d.natf = natf;
        auto req = [](int r) { if (r != EMAGLS_OK) throw Error(r, g_last_error); };
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(g_atfsets_mu);
        const size_t out_bytes = sizeof(double) * (size_t)filter_len * (size_t)nmics;
        for (int64_t first = 0; first < nsets;) {
            const int n = (int)std::min<int64_t>(SWEEP_MULTI_MAX, nsets - first);
            const int slot = n == SWEEP_MULTI_MAX ? 0 : 1;
            SetsCache* c = &g_atfsets[slot];
            if (!(c->n == n && c->device == dev && same_desc(c->desc, d))) {
                c->release();
                try {
                    for (int j = 0; j < n; ++j) {
                        emagls_plan* p = nullptr;
                        req(emagls_plan_create(&d, &p));
                        c->plans.push_back(p);
                    }
                    if (n > 1) {
                        g_batch_max_override = SWEEP_MULTI_MAX;
                        const int r = emagls_batch_create(c->plans.data(), n, &c->batch);
                        g_batch_max_override = 0;
                        req(r);
                    }
                } catch (...) { g_batch_max_override = 0; c->release(); throw; }
                c->desc = d; c->device = dev; c->n = n;
            }
            try {
                // the ATF set and the grids: host -> plan 0, plan 0 -> the others on the device (the set is 268 MB at config 5)
                emagls_plan* p0 = c->plans[0];
                req(emagls_plan_set_hrir_grid(p0, hrir_azi, hrir_zen));
                req(emagls_plan_set_atfs(p0, atf_irs, atf_azi, atf_zen));
                for (int j = 1; j < n; ++j) {
                    emagls_plan* p = c->plans[(size_t)j];
                    for (const char* name : {"atf", "atf_azi", "atf_zen", "hrir_azi", "hrir_zen"})
                        HIP_CHECK(hipMemcpyAsync(p->get(name), p0->get(name), p0->bufs[name].bytes, hipMemcpyDeviceToDevice, p0->stream));
                    p->have_atfs = true; p->have_hrir_grid = true;
                    ++p->atf_side_version;
                }
                HIP_CHECK(hipStreamSynchronize(p0->stream));
                for (int j = 0; j < n; ++j)
                    req(emagls_plan_set_hrirs(c->plans[(size_t)j], hL + (first + j) * nsamp * ndirs, hR + (first + j) * nsamp * ndirs));
                if (n == 1) {
                    req(emagls_plan_execute(p0));
                    req(emagls_plan_get_filters(p0, (char*)wL + first * out_bytes, (char*)wR + first * out_bytes));
                } else {
                    std::vector<void*> pl((size_t)n), pr((size_t)n);
                    for (int j = 0; j < n; ++j) { pl[(size_t)j] = (char*)wL + (first + j) * out_bytes; pr[(size_t)j] = (char*)wR + (first + j) * out_bytes; }
                    req(emagls_batch_execute(c->batch));
                    req(emagls_batch_get_filters(c->batch, pl.data(), pr.data()));
                }
                if (mean_dev) {
                    emagls_plan_info info;
                    req(emagls_plan_get_info(p0, &info));
                    *mean_dev = info.mean_grid_dev_deg;
                }
            } catch (...) { c->release(); throw; }
            first += n;
        }
    });
}

int emagls_binaural_decode(const double* in, int64_t nsamp, int64_t nch, const double* wL, const double* wR, int64_t len,
                           int compensate_delay, double* out) {
    return decode_entry(in, false, nsamp, nch, wL, wR, false, len, compensate_delay, out, nullptr);
}
int emagls_binaural_decode_complex(const void* in, int in_is_complex, int64_t nsamp, int64_t nch, const void* wL, const void* wR,
                                   int filters_are_complex, int64_t len, int compensate_delay, double* out, double* imag_abs_sum) {
    return decode_entry(in, in_is_complex != 0, nsamp, nch, wL, wR, filters_are_complex != 0, len, compensate_delay, out, imag_abs_sum);
}

}  // extern "C"
