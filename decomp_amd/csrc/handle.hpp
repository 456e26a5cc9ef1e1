// Opaque library handle: device binding, stream, grow-only device workspace and
// pinned host scalars for read-backs.  One handle per process/device; calls on one
// handle are not re-entrant.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/decomp_hip.h"

struct dcp_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    // The workspace arena is shared by every arena-using call on this handle.  arena_stream is the stream
    // the last such call enqueued on; when the next one runs on a different stream (dcp_set_stream in
    // between), ws_reserve makes the new stream wait for the old one with this event -- calls that never
    // touch the arena (the row movers of the out-of-core feed) are not ordered against anything.
    hipEvent_t stream_switch = nullptr;
    hipStream_t arena_stream = nullptr;
    bool arena_stream_set = false;
    // Side stream for work that can run BESIDE a kernel that leaves most of the chip idle (the atom
    // sweep's look-ahead product next to the one-workgroup recursion); created on first use and
    // ordered against the main stream with events only (side_after_main / main_after_side).
    hipStream_t side = nullptr;
    hipEvent_t ev_main = nullptr, ev_side = nullptr;
    // Workspace: one grow-only arena.  A public call plans its total need, reserves it
    // (ws_reserve: reallocates only when the plan outgrows the arena, i.e. on the first
    // call of a given problem size, never in steady state), then bumps (ws_alloc).
    char* arena = nullptr;
    size_t arena_bytes = 0;
    size_t arena_used = 0;
    // pinned host scratch for scalar read-backs
    void* host_pinned = nullptr;
    size_t host_pinned_bytes = 0;
    std::string err;
    // RCCL communicator of the sample-sharded solvers (comm.hip; ncclComm_t kept opaque here)
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    // ... or a caller-supplied exchange (dcp_comm_set_external: MPI, gloo, a test double)
    dcp_allreduce_fn comm_ext = nullptr;
    void* comm_ext_user = nullptr;
    // a row gather registered by dcp_dict_prefetch_rows_bytes: the next dictionary step runs it on the side
    // stream beside its atom sweep (rows of the NEXT minibatch while the chip is nearly idle)
    const void* pf_in = nullptr;
    const int64_t* pf_index = nullptr;
    void* pf_out = nullptr;
    int64_t pf_rows = 0, pf_row_bytes = 0;
    bool pf_inflight = false;   // started on the side stream, not yet joined
    // coordinate descent inside the dictionary step: the stop flag of the solve's last ten sweeps is still on its way
    // to this pinned word when lasso_solve returns; lasso_settle_deferred() turns it into *lasso_deferred_it
    int* lasso_deferred_flag = nullptr;
    int* lasso_deferred_it = nullptr;
    int lasso_deferred_it_met = 0;
    int lasso_it_sink = 0;
    // parallel_cd inside the dictionary step: the caller-supplied shuffle table (dcp_dict_set_pcd_order)
    const int* pcd_order = nullptr;
    int64_t pcd_rows = 0, pcd_K = 0;
    // optional per-kernel timing (dcp_profile_*): hipEvent pairs around labelled launches
    bool prof_on = false;
    unsigned prof_mask = 0xffffffffu;   // labels that are timed while prof_on
    struct ProfRec { int label; hipEvent_t a, b; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[DCP_PROF_NLABELS] = {0};
    long long prof_cnt[DCP_PROF_NLABELS] = {0};
};

namespace dcp {

inline int fail(dcp_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

#define DCP_HIP_OK(h, expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return dcp::fail((h), DCP_ERR_HIP,                                               \
                             std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

#define DCP_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != DCP_OK) return _rc; \
    } while (0)

// Times everything enqueued on the handle's stream during its lifetime under `label`
// (only when profiling is on; otherwise free).
struct ProfScope {
    dcp_handle* h;
    int label;
    hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t take(dcp_handle* h) {
        if (!h->prof_pool.empty()) {
            hipEvent_t e = h->prof_pool.back();
            h->prof_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    ProfScope(dcp_handle* h_, int label_) : h(h_), label(label_) {
        if (!h->prof_on || !((h->prof_mask >> label) & 1u)) return;
        a = take(h);
        b = take(h);
        if (a) (void)hipEventRecord(a, h->stream);
    }
    ~ProfScope() {
        if (!a || !b) return;
        (void)hipEventRecord(b, h->stream);
        h->prof_recs.push_back({label, a, b});
    }
};

// Events that only order the handle's two streams on ONE device: no timing, and no system-scope fence when they are
// recorded (the default's cache write-back idles the recording stream ~6 us; nothing on the host reads behind them).
constexpr unsigned kOrderingEvent = hipEventDisableTiming | hipEventDisableSystemFence;

// what is enqueued on h->side after this call starts after the work already on h->stream
inline int side_after_main(dcp_handle* h) {
    if (h->side == nullptr) {
        if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_main, kOrderingEvent) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_side, kOrderingEvent) != hipSuccess)
            return fail(h, DCP_ERR_HIP, "side stream creation failed");
    }
    if (hipEventRecord(h->ev_main, h->stream) != hipSuccess ||
        hipStreamWaitEvent(h->side, h->ev_main, 0) != hipSuccess)
        return fail(h, DCP_ERR_HIP, "side stream fork failed");
    return DCP_OK;
}
// what is enqueued on h->stream after this call starts after the work already on h->side
inline int main_after_side(dcp_handle* h) {
    if (hipEventRecord(h->ev_side, h->side) != hipSuccess ||
        hipStreamWaitEvent(h->stream, h->ev_side, 0) != hipSuccess)
        return fail(h, DCP_ERR_HIP, "side stream join failed");
    return DCP_OK;
}

// gather / scatter rows on `stream` (util.hip; the row movers of the minibatch containers)
int move_rows_on(dcp_handle* h, hipStream_t stream, const void* in, const int64_t* in_index, void* out,
                 const int64_t* out_index, int64_t rows, int64_t row_bytes);

// Starts the handle's registered row gather on the side stream, ordered after what is on the main stream now.
inline int start_registered_prefetch(dcp_handle* h) {
    if (h->pf_rows <= 0 || h->pf_row_bytes <= 0) return DCP_OK;
    DCP_TRY(side_after_main(h));
    DCP_TRY(move_rows_on(h, h->side, h->pf_in, h->pf_index, h->pf_out, nullptr, h->pf_rows, h->pf_row_bytes));
    h->pf_rows = 0;
    h->pf_inflight = true;
    return DCP_OK;
}

inline void ws_reset(dcp_handle* h) {
    h->arena_used = 0;
}

// The arena's previous user may still be running on another stream: the current stream waits for it
// (an event recorded on the previous stream, no host sync).  decomp_hip.h makes the caller keep that stream
// alive until this point; the device synchronisation below only covers a FAILED record / wait on a live
// stream (it is not a defence against a destroyed one: that is a use-after-free inside the runtime).
inline void ws_order_streams(dcp_handle* h) {
    if (h->arena_stream_set && h->arena_stream != h->stream && h->arena != nullptr) {
        bool ok = false;
        if (h->stream_switch == nullptr)
            (void)hipEventCreateWithFlags(&h->stream_switch, kOrderingEvent);
        if (h->stream_switch != nullptr && hipEventRecord(h->stream_switch, h->arena_stream) == hipSuccess &&
            hipStreamWaitEvent(h->stream, h->stream_switch, 0) == hipSuccess)
            ok = true;
        if (!ok) {
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
        }
    }
    h->arena_stream = h->stream;
    h->arena_stream_set = true;
}

// Reserve the total workspace for a call up front (so no allocation happens
// between kernels).  Returns DCP_OK or an error code.  Every arena-using entry point calls this first.
inline int ws_reserve(dcp_handle* h, size_t bytes) {
    ws_order_streams(h);
    if (bytes <= h->arena_bytes) return DCP_OK;
    if (h->arena) {
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return fail(h, DCP_ERR_HIP, hipGetErrorString(e));
        (void)hipFree(h->arena);
        h->arena = nullptr;
        h->arena_bytes = 0;
    }
    size_t want = bytes + (bytes >> 3) + (1u << 20);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        want = bytes;
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess)
        return fail(h, DCP_ERR_NOMEM,
                    "workspace hipMalloc of " + std::to_string(bytes) + " bytes failed: " +
                        hipGetErrorString(e));
    h->arena = static_cast<char*>(p);
    h->arena_bytes = want;
    h->arena_used = 0;
    return DCP_OK;
}

inline size_t align256(size_t b) { return (b + 255) & ~size_t(255); }

// Bump allocation from the reserved arena; nullptr when the reservation was too small
// (a library bug, reported by the caller as DCP_ERR_INTERNAL).
template <class T>
inline T* ws_alloc(dcp_handle* h, size_t count) {
    const size_t bytes = align256(count * sizeof(T));
    if (h->arena_used + bytes > h->arena_bytes) return nullptr;
    T* p = reinterpret_cast<T*>(h->arena + h->arena_used);
    h->arena_used += bytes;
    return p;
}

// Sizing helper mirroring ws_alloc.
struct WsPlan {
    size_t total = 0;
    template <class T>
    void add(size_t count) {
        total += align256(count * sizeof(T));
    }
};

inline int host_scratch(dcp_handle* h, size_t bytes, void** out) {
    if (bytes > h->host_pinned_bytes) {
        if (h->host_pinned) (void)hipHostFree(h->host_pinned);
        h->host_pinned = nullptr;
        h->host_pinned_bytes = 0;
        size_t want = bytes < 4096 ? 4096 : bytes;
        hipError_t e = hipHostMalloc(&h->host_pinned, want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(h, DCP_ERR_NOMEM, "hipHostMalloc failed");
        h->host_pinned_bytes = want;
    }
    *out = h->host_pinned;
    return DCP_OK;
}

}  // namespace dcp
