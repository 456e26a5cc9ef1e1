// C ABI: NMF multiplicative update (see include/decomp_hip.h for the contract and the
// reference lines each entry point replaces).
#include "comm.hpp"
#include "nmf_impl.hpp"

using namespace dcp;

namespace {

template <class T>
int check_nmf_args(dcp_handle* h, const T* Y, const T* X, const T* D, int64_t N, int64_t F,
                   int64_t K, int lik) {
    if (!h) return DCP_ERR_INVALID;
    if (!Y || !X || !D) return fail(h, DCP_ERR_INVALID, "null array pointer");
    if (N <= 0 || F <= 0 || K <= 0) return fail(h, DCP_ERR_INVALID, "sizes must be positive");
    if (N > 0x7fffffffLL || F > 0x7fffffffLL || K > 0x7fffffffLL || F + K > 0x3fffffffLL)
        return fail(h, DCP_ERR_INVALID, "dimension exceeds 2^31-1");
    if (lik != DCP_LIK_L2 && lik != DCP_LIK_KL) return fail(h, DCP_ERR_INVALID, "bad likelihood");
    return DCP_OK;
}

template <class T>
int nmf_mu_solve(dcp_handle* h, const T* Y, const T* mask, T* X, T* D, int64_t N, int64_t F,
                 int64_t K, int lik, T tol, int maxiter, int* it_out, T* last_maxdiff,
                 T* resid_trace, bool sharded = false) {
    DCP_TRY(check_nmf_args(h, Y, X, D, N, F, K, lik));
    if (!it_out) return fail(h, DCP_ERR_INVALID, "it_out is null");
    if (sharded && !comm_active(h))
        return fail(h, DCP_ERR_COMM, "dcp_nmf_mu_sharded_* needs a communicator (dcp_comm_init)");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    const bool masked = mask != nullptr;
    NmfShape<T> s{N, F, K, lik, masked};
    const int64_t W = nmf_stats_width(F, K, lik, masked);
    const bool want_resid = resid_trace != nullptr;
    const bool gram = (lik == DCP_LIK_L2 && !masked);
    const int resid_blocks = 1024;

    WsPlan plan;
    nmf_plan_stats(plan, s, masked);
    nmf_plan_update<T>(plan, F, K);
    plan.add<T>((size_t)K * W);   // stats
    plan.add<T>((size_t)K * F);   // second D buffer
    plan.add<T>((size_t)N * K);   // second x buffer
    plan.add<T>(2);               // max|dD| of the two iterations in flight
    plan.add<unsigned int>(4);    // arrival ticket of the normalisation's workgroups
    const bool want_bits = masked && std::is_same<T, float>::value && lik == DCP_LIK_L2;
    if (want_bits) {
        plan.add<uint32_t>(mask_bits_words(N, F));
        plan.add<int>(4);
    }
    if (want_resid) {
        if (gram) plan.add<T>((size_t)N * F);
        plan.add<double>(resid_blocks);
    }
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    NmfStatsWs<T> ws;
    NmfUpdateWs<T> wu;
    DCP_TRY(nmf_carve_stats(h, ws, s, masked));
    DCP_TRY(nmf_carve_update(h, wu, F, K));
    T* stats = ws_alloc<T>(h, (size_t)K * W);
    T* D2 = ws_alloc<T>(h, (size_t)K * F);
    T* X2 = ws_alloc<T>(h, (size_t)N * K);
    T* maxdiff_dev = ws_alloc<T>(h, 2);
    unsigned int* ticket = ws_alloc<unsigned int>(h, 4);
    uint32_t* mbits = nullptr;
    int* mflag = nullptr;
    if (want_bits) {
        mbits = ws_alloc<uint32_t>(h, mask_bits_words(N, F));
        mflag = ws_alloc<int>(h, 4);
        if (!mbits || !mflag) return fail(h, DCP_ERR_INTERNAL, "nmf workspace plan mismatch");
    }
    T* resid_tmp = nullptr;
    double* resid_part = nullptr;
    if (want_resid) {
        resid_tmp = gram ? ws_alloc<T>(h, (size_t)N * F) : ws.f;
        resid_part = ws_alloc<double>(h, resid_blocks);
    }
    if (!stats || !D2 || !X2 || !maxdiff_dev || !ticket || (want_resid && (!resid_tmp || !resid_part)))
        return fail(h, DCP_ERR_INTERNAL, "nmf workspace plan mismatch");
    void* hostv = nullptr;
    DCP_TRY(host_scratch(h, sizeof(double) * (resid_blocks + 4), &hostv));
    T* host_md = reinterpret_cast<T*>(hostv);             // [2]
    double* host_part = reinterpret_cast<double*>(hostv) + 2;
    // The stop test polls the pinned word the normalisation's last workgroup stores max|dD| into (a sentinel of -1
    // is put there before the iteration is enqueued; max|dD| >= 0 or NaN).  No event in the loop: the barrier packet of
    // a hipEventRecord (system-scope release) cost ~6 us of idle GPU per iteration behind the normalisation.
    auto wait_md = [&](int slot, T* out) -> int {
        volatile T* v = host_md + slot;
        bool seen = false;
        for (long spin = 0; spin < 400000000L; ++spin) {
            if (!(*v == T(-1))) { seen = true; break; }
            __builtin_ia32_pause();
        }
        if (!seen) DCP_HIP_OK(h, hipStreamSynchronize(h->stream));   // ~2 s of polling: fall back to a blocking wait
        *out = *v;
        return DCP_OK;
    };

    const T* Ypre = Y;
    if (masked) {  // y * mask is loop invariant (grads.py:114,124 recompute it every call)
        int binary = 0;
        DCP_TRY(nmf_mask_prepare<T>(h, Y, mask, N, F, ws.Ym, mbits, mflag, &binary));
        Ypre = ws.Ym;
        if (binary) ws.mbits = mbits;
    }
    DCP_HIP_OK(h, hipMemsetAsync(maxdiff_dev, 0, 2 * sizeof(T), h->stream));
    DCP_HIP_OK(h, hipMemsetAsync(ticket, 0, 4 * sizeof(unsigned int), h->stream));

    // Iteration `it` reads (x_{it-1}, D_{it-1}) from (Xc, Dc) and writes (x_it, D_it) to (Xn, Dn);
    // its max|dD| lands in host slot it&1.  The stop test of iteration it-1
    // (batch_mu.py:22) is evaluated AFTER iteration it has been enqueued, so the GPU never
    // idles on the host; when it-1 turns out to have converged, iteration it is discarded:
    // its inputs (Xc, Dc) are exactly the state the reference returns.
    T* Xc = X;  T* Xn = X2;
    T* Dc = D;  T* Dn = D2;
    int result_it = maxiter;   // batch_mu.py:26
    T md_last = T(0);
    bool converged = false;
    for (int it = 1; it < maxiter; ++it) {  // batch_mu.py:16
        const int slot = it & 1;
        DCP_TRY(nmf_stats<T>(h, Ypre, mask, Xc, Xn, Dc, s, stats, ws));
        if (sharded) {   // the one exchange of the step: sums over rows become sums over ranks
            ProfScope ps(h, DCP_PROF_EXCHANGE);
            DCP_TRY(comm_allreduce_sum(h, stats, (size_t)K * W,
                                       std::is_same<T, float>::value ? COMM_F32 : COMM_F64));
        }
        // max|D - D_new| reaches the host without a copy kernel: the normalisation's last-arriving workgroup stores
        // it into the pinned (device-mapped) slot, which the stop test polls
        *reinterpret_cast<volatile T*>(host_md + slot) = T(-1);
        DCP_TRY(nmf_update<T>(h, stats, Dc, Dn, F, K, lik, masked, maxdiff_dev + slot, wu,
                              maxdiff_dev + (slot ^ 1), ticket, host_md + slot));
        if (want_resid) {   // parity/debug mode: synchronous
            DCP_TRY(nmf_residual<T>(h, Y, mask, Xn, Dn, N, F, K, resid_tmp, resid_part,
                                    resid_blocks));
            DCP_HIP_OK(h, hipMemcpyAsync(host_part, resid_part, sizeof(double) * resid_blocks,
                                         hipMemcpyDeviceToHost, h->stream));
            DCP_HIP_OK(h, hipStreamSynchronize(h->stream));
            double acc = 0.0;
            for (int i = 0; i < resid_blocks; ++i) acc += host_part[i];
            resid_trace[it - 1] = (T)sqrt(acc);
        }
        if (it > 1) {   // stop test of the PREVIOUS iteration
            DCP_TRY(wait_md(slot ^ 1, &md_last));
            if (md_last < tol) {   // a NaN compares false, as in NumPy
                result_it = it - 1;
                converged = true;
                break;             // (Xc, Dc) hold x_{it-1} and D_new of iteration it-1
            }
        }
        T* t = Xc; Xc = Xn; Xn = t;
        t = Dc; Dc = Dn; Dn = t;
    }
    if (!converged && maxiter > 1) {   // stop test of the last iteration
        const int slot = (maxiter - 1) & 1;
        DCP_TRY(wait_md(slot, &md_last));
        if (md_last < tol) result_it = maxiter - 1;
    }
    DCP_HIP_OK(h, hipStreamSynchronize(h->stream));   // drain (incl. a discarded iteration)
    if (Xc != X)
        DCP_HIP_OK(h, hipMemcpyAsync(X, Xc, sizeof(T) * (size_t)N * K, hipMemcpyDeviceToDevice,
                                     h->stream));
    if (Dc != D)
        DCP_HIP_OK(h, hipMemcpyAsync(D, Dc, sizeof(T) * (size_t)K * F, hipMemcpyDeviceToDevice,
                                     h->stream));
    DCP_HIP_OK(h, hipStreamSynchronize(h->stream));
    *it_out = result_it;
    if (last_maxdiff) *last_maxdiff = md_last;
    return DCP_OK;
}

template <class T>
int nmf_mu_stats_api(dcp_handle* h, const T* Y, const T* mask, const T* X, T* X_out, const T* D,
                     int64_t N, int64_t F, int64_t K, int lik, T* stats) {
    DCP_TRY(check_nmf_args(h, Y, X, D, N, F, K, lik));
    if (!stats || !X_out) return fail(h, DCP_ERR_INVALID, "stats / X_out is null");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    const bool masked = mask != nullptr;
    NmfShape<T> s{N, F, K, lik, masked};
    WsPlan plan;
    nmf_plan_stats(plan, s, masked);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    NmfStatsWs<T> ws;
    DCP_TRY(nmf_carve_stats(h, ws, s, masked));
    const T* Ypre = Y;
    if (masked) {
        hipLaunchKernelGGL((mul_mask_kernel<T>), dim3(grid_for(N * F)), dim3(256), 0, h->stream, Y,
                           mask, (long)N, (long)F, (long)F, ws.Ym);
        DCP_HIP_OK(h, hipGetLastError());
        Ypre = ws.Ym;
    }
    return nmf_stats<T>(h, Ypre, mask, X, X_out, D, s, stats, ws);
}

// dcp_nmf_mu_stats_* with the loop-invariant mask work done once by dcp_nmf_mask_prepare_*.
template <class T>
int nmf_mu_stats_prepared_api(dcp_handle* h, const T* Ym, const T* mask, const uint32_t* bits, const T* X,
                              T* X_out, const T* D, int64_t N, int64_t F, int64_t K, int lik, T* stats) {
    DCP_TRY(check_nmf_args(h, Ym, X, D, N, F, K, lik));
    if (!stats || !X_out || !mask) return fail(h, DCP_ERR_INVALID, "stats / X_out / mask is null");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    NmfShape<T> s{N, F, K, lik, true};
    WsPlan plan;
    nmf_plan_stats(plan, s, false);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    NmfStatsWs<T> ws;
    DCP_TRY(nmf_carve_stats(h, ws, s, false));
    ws.mbits = bits;
    return nmf_stats<T>(h, Ym, mask, X, X_out, D, s, stats, ws);
}

template <class T>
int nmf_mask_prepare_api(dcp_handle* h, const T* Y, const T* mask, int64_t N, int64_t F, T* Ym,
                         uint32_t* bits, int* binary) {
    if (!h) return DCP_ERR_INVALID;
    if (!Y || !mask || !Ym || !binary) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (N <= 0 || F <= 0) return fail(h, DCP_ERR_INVALID, "sizes must be positive");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    WsPlan plan;
    plan.add<int>(4);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    int* flag = ws_alloc<int>(h, 4);
    if (!flag) return fail(h, DCP_ERR_INTERNAL, "workspace plan mismatch");
    return nmf_mask_prepare<T>(h, Y, mask, N, F, Ym, bits, flag, binary);
}

template <class T>
int nmf_mu_update_api(dcp_handle* h, const T* stats, const T* D, T* D_new, int64_t F, int64_t K,
                      int lik, int masked, T* maxdiff_dev, T* maxdiff_next) {
    if (!h) return DCP_ERR_INVALID;
    if (!stats || !D || !D_new || !maxdiff_dev) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (F <= 0 || K <= 0) return fail(h, DCP_ERR_INVALID, "sizes must be positive");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    // NOTE: shares the arena with dcp_nmf_mu_stats_*: the stats call's temporaries are dead
    // by now (same stream), `stats` itself is caller memory.
    WsPlan plan;
    nmf_plan_update<T>(plan, F, K);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    NmfUpdateWs<T> wu;
    DCP_TRY(nmf_carve_update(h, wu, F, K));
    return nmf_update<T>(h, stats, D, D_new, F, K, lik, masked != 0, maxdiff_dev, wu, maxdiff_next);
}

template <class T>
int nmf_residual_api(dcp_handle* h, const T* Y, const T* mask, const T* X, const T* D, int64_t N,
                     int64_t F, int64_t K, double* out) {
    DCP_TRY(check_nmf_args(h, Y, X, D, N, F, K, DCP_LIK_L2));
    if (!out) return fail(h, DCP_ERR_INVALID, "out is null");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    const int blocks = 1024;
    WsPlan plan;
    plan.add<T>((size_t)N * F);
    plan.add<double>(blocks);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    T* tmp = ws_alloc<T>(h, (size_t)N * F);
    double* part = ws_alloc<double>(h, blocks);
    if (!tmp || !part) return fail(h, DCP_ERR_INTERNAL, "workspace plan mismatch");
    void* hostv = nullptr;
    DCP_TRY(host_scratch(h, sizeof(double) * blocks, &hostv));
    DCP_TRY(nmf_residual<T>(h, Y, mask, X, D, N, F, K, tmp, part, blocks));
    DCP_HIP_OK(h, hipMemcpyAsync(hostv, part, sizeof(double) * blocks, hipMemcpyDeviceToHost,
                                 h->stream));
    DCP_HIP_OK(h, hipStreamSynchronize(h->stream));
    double acc = 0.0;
    for (int i = 0; i < blocks; ++i) acc += reinterpret_cast<double*>(hostv)[i];
    *out = sqrt(acc);
    return DCP_OK;
}

// grads.py:108-125 / 143-160 on a minibatch: optional x update(s), then the two parts of the
// D gradient as explicit [K,F] arrays (what serizel.py / kasai.py accumulate and mix).
template <class T>
int nmf_grads_api(dcp_handle* h, const T* Y, const T* mask, T* X, const T* D, int64_t N, int64_t F,
                  int64_t K, int lik, int n_x_updates, T* grad_pos, T* grad_neg) {
    DCP_TRY(check_nmf_args(h, Y, X, D, N, F, K, lik));
    if (!grad_pos || !grad_neg) return fail(h, DCP_ERR_INVALID, "null gradient pointer");
    if (n_x_updates < 0) return fail(h, DCP_ERR_INVALID, "negative update count");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    const bool masked = mask != nullptr;
    const bool gram = (lik == DCP_LIK_L2 && !masked);
    NmfShape<T> s{N, F, K, lik, masked};
    const int64_t W = nmf_stats_width(F, K, lik, masked);
    WsPlan plan;
    nmf_plan_stats(plan, s, masked);
    plan.add<T>((size_t)K * W);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    NmfStatsWs<T> ws;
    DCP_TRY(nmf_carve_stats(h, ws, s, masked));
    T* stats = ws_alloc<T>(h, (size_t)K * W);
    if (!stats) return fail(h, DCP_ERR_INTERNAL, "nmf workspace plan mismatch");
    const T* Ypre = Y;
    if (masked) {
        hipLaunchKernelGGL((mul_mask_kernel<T>), dim3(grid_for(N * F)), dim3(256), 0, h->stream, Y,
                           mask, (long)N, (long)F, (long)F, ws.Ym);
        DCP_HIP_OK(h, hipGetLastError());
        Ypre = ws.Ym;
    }
    for (int i = 0; i < n_x_updates; ++i)
        DCP_TRY(nmf_stats<T>(h, Ypre, mask, X, X, D, s, stats, ws, 1));
    DCP_TRY(nmf_stats<T>(h, Ypre, mask, X, X, D, s, stats, ws, 2));
    // stats -> (pos, neg)
    DCP_HIP_OK(h, hipMemcpy2DAsync(grad_pos, sizeof(T) * F, stats, sizeof(T) * W, sizeof(T) * F, K,
                                   hipMemcpyDeviceToDevice, h->stream));
    if (gram) {   // neg = (x^T x) D
        GemmArgs<T> a;
        a.A = stats + F; a.lda = W; a.B = D; a.ldb = F; a.M = (int)K; a.N = (int)F; a.K = (int)K;
        a.tile = TILE_SMALL;
        DCP_LAUNCH_OK(h, (gemm<FORM_NN>(h->stream, a, EpiStore<T>{grad_neg, (long)F})));
    } else {
        DCP_HIP_OK(h, hipMemcpy2DAsync(grad_neg, sizeof(T) * F, stats + F, sizeof(T) * W,
                                       sizeof(T) * F, K, hipMemcpyDeviceToDevice, h->stream));
    }
    return DCP_OK;
}

// Gaussian / Poisson.grad_x (grads.py:108-115, 143-150): the two parts of the x gradient, [N, K] each.
template <class T>
int nmf_grad_x_api(dcp_handle* h, const T* Y, const T* mask, const T* X, const T* D, int64_t N, int64_t F,
                   int64_t K, int lik, T* grad_pos, T* grad_neg) {
    DCP_TRY(check_nmf_args(h, Y, X, D, N, F, K, lik));
    if (!grad_pos || !grad_neg) return fail(h, DCP_ERR_INVALID, "null gradient pointer");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    const bool masked = mask != nullptr;
    NmfShape<T> s{N, F, K, lik, masked};
    WsPlan plan;
    nmf_plan_stats(plan, s, masked);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    NmfStatsWs<T> ws;
    DCP_TRY(nmf_carve_stats(h, ws, s, masked));
    const T* Ypre = Y;
    if (masked) {
        hipLaunchKernelGGL((mul_mask_kernel<T>), dim3(grid_for(N * F)), dim3(256), 0, h->stream, Y, mask,
                           (long)N, (long)F, (long)F, ws.Ym);
        DCP_HIP_OK(h, hipGetLastError());
        Ypre = ws.Ym;
    }
    return nmf_grad_x<T>(h, Ypre, mask, X, D, s, grad_pos, grad_neg, ws);
}

// Gaussian.logp (grads.py:127-135): sum((-0.5 ((y - x d) / scale)^2 - log(scale) - pi * 0.5) [* mask]).
template <class T>
int nmf_gauss_logp_api(dcp_handle* h, const T* Y, const T* mask, const T* X, const T* D, int64_t N,
                       int64_t F, int64_t K, double scale, double* out) {
    DCP_TRY(check_nmf_args(h, Y, X, D, N, F, K, DCP_LIK_L2));
    if (!out) return fail(h, DCP_ERR_INVALID, "out is null");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    const int blocks = 1024;
    WsPlan plan;
    plan.add<T>((size_t)N * F);
    plan.add<double>(blocks);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    T* tmp = ws_alloc<T>(h, (size_t)N * F);
    double* part = ws_alloc<double>(h, blocks);
    if (!tmp || !part) return fail(h, DCP_ERR_INTERNAL, "workspace plan mismatch");
    void* hostv = nullptr;
    DCP_TRY(host_scratch(h, sizeof(double) * blocks, &hostv));
    GemmArgs<T> a;
    a.A = X; a.lda = K; a.B = D; a.ldb = F; a.M = (int)N; a.N = (int)F; a.K = (int)K;
    DCP_LAUNCH_OK(h, (gemm<FORM_NN>(h->stream, a, EpiResidual<T>{Y, F, nullptr, 0, tmp, F})));   // d = y - x D
    hipLaunchKernelGGL((gauss_logp_partial_kernel<T>), dim3(blocks), dim3(256), 0, h->stream, (const T*)tmp,
                       mask, (long)N * F, 1.0 / scale, log(scale) + 3.14159265358979323846 * 0.5, part);
    DCP_LAUNCH_OK(h, hipGetLastError());
    DCP_HIP_OK(h, hipMemcpyAsync(hostv, part, sizeof(double) * blocks, hipMemcpyDeviceToHost, h->stream));
    DCP_HIP_OK(h, hipStreamSynchronize(h->stream));
    double acc = 0.0;
    for (int i = 0; i < blocks; ++i) acc += reinterpret_cast<double*>(hostv)[i];
    *out = acc;
    return DCP_OK;
}

// D_new = l2_strict(rule(D, P, Q)) and max|D - D_new| (host):
//   alpha < 0 : D * max(P,0) / max(Q,eps)                         (grads.py:93)
//   alpha >= 0: max(D * ((1-alpha) + alpha * P / max(Q,eps)), 0)  (kasai.py:77-78)
template <class T>
__global__ void __launch_bounds__(256) nmf_rule_kernel(const T* __restrict__ D, const T* __restrict__ P,
                                                       const T* __restrict__ Q, long n, T alpha,
                                                       T* __restrict__ U) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const T q = max_np(Q[i], T(1.0e-15));
        T u;
        if (alpha < T(0)) {
            u = D[i] * max_np(P[i], T(0)) / q;
        } else {
            u = D[i] * ((T(1) - alpha) + alpha * P[i] / q);
            u = max_np(u, T(0));
        }
        U[i] = u;
    }
}

template <class T>
__global__ void __launch_bounds__(256) axpby_kernel(long n, T a, const T* x, T b, T* y) {
    // a zero coefficient means "do not read that operand" (BLAS semantics): y may be
    // uninitialised when b == 0, and 0 * NaN / 0 * Inf must not leak into the result.
    // x and y may alias (no __restrict__).
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        T v;
        if (b == T(0)) v = a * x[i];
        else if (a == T(0)) v = b * y[i];
        else v = a * x[i] + b * y[i];
        y[i] = v;
    }
}

template <class T>
int nmf_apply_api(dcp_handle* h, const T* D, const T* P, const T* Q, double alpha, T* D_new, int64_t K,
                  int64_t F, double* maxdiff) {
    if (!h) return DCP_ERR_INVALID;
    if (!D || !P || !Q || !D_new || !maxdiff) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (K <= 0 || F <= 0) return fail(h, DCP_ERR_INVALID, "sizes must be positive");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    WsPlan plan;
    nmf_plan_update<T>(plan, F, K);
    plan.add<T>(2);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    NmfUpdateWs<T> wu;
    DCP_TRY(nmf_carve_update(h, wu, F, K));
    T* md = ws_alloc<T>(h, 2);
    if (!md) return fail(h, DCP_ERR_INTERNAL, "workspace plan mismatch");
    hipLaunchKernelGGL((nmf_rule_kernel<T>), dim3(grid_for(K * F)), dim3(256), 0, h->stream, D, P, Q,
                       (long)(K * F), (T)alpha, wu.U);
    DCP_HIP_OK(h, hipGetLastError());
    hipLaunchKernelGGL((row_normalize_kernel<T>), dim3((unsigned)K), dim3(256), 0, h->stream,
                       (const T*)wu.U, (long)F, (long)F, 1, D, (long)F, D_new, (long)F, wu.rowmax,
                       (T*)nullptr, (T*)nullptr, (T*)nullptr);
    DCP_HIP_OK(h, hipGetLastError());
    hipLaunchKernelGGL((final_max_kernel<T>), dim3(1), dim3(256), 0, h->stream, (const T*)wu.rowmax,
                       (long)K, md);
    DCP_HIP_OK(h, hipGetLastError());
    void* hostv = nullptr;
    DCP_TRY(host_scratch(h, 64, &hostv));
    DCP_HIP_OK(h, hipMemcpyAsync(hostv, md, sizeof(T), hipMemcpyDeviceToHost, h->stream));
    DCP_HIP_OK(h, hipStreamSynchronize(h->stream));
    *maxdiff = (double)(*reinterpret_cast<T*>(hostv));
    return DCP_OK;
}

template <class T>
int axpby_api(dcp_handle* h, int64_t n, double a, const T* x, double b, T* y) {
    if (!h) return DCP_ERR_INVALID;
    if (!x || !y) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (n < 0) return fail(h, DCP_ERR_INVALID, "negative size");
    if (n == 0) return DCP_OK;
    DCP_HIP_OK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL((axpby_kernel<T>), dim3(grid_for(n)), dim3(256), 0, h->stream, (long)n, (T)a, x,
                       (T)b, y);
    DCP_HIP_OK(h, hipGetLastError());
    return DCP_OK;
}

}  // namespace

extern "C" {

int dcp_nmf_grads_f32(dcp_handle* h, const float* Y, const float* mask, float* X, const float* D,
                      int64_t N, int64_t F, int64_t K, int likelihood, int n_x_updates,
                      float* grad_pos, float* grad_neg) {
    return nmf_grads_api<float>(h, Y, mask, X, D, N, F, K, likelihood, n_x_updates, grad_pos, grad_neg);
}
int dcp_nmf_grads_f64(dcp_handle* h, const double* Y, const double* mask, double* X, const double* D,
                      int64_t N, int64_t F, int64_t K, int likelihood, int n_x_updates,
                      double* grad_pos, double* grad_neg) {
    return nmf_grads_api<double>(h, Y, mask, X, D, N, F, K, likelihood, n_x_updates, grad_pos, grad_neg);
}
int dcp_nmf_grad_x_f32(dcp_handle* h, const float* Y, const float* mask, const float* X, const float* D,
                       int64_t N, int64_t F, int64_t K, int likelihood, float* grad_pos, float* grad_neg) {
    return nmf_grad_x_api<float>(h, Y, mask, X, D, N, F, K, likelihood, grad_pos, grad_neg);
}
int dcp_nmf_grad_x_f64(dcp_handle* h, const double* Y, const double* mask, const double* X, const double* D,
                       int64_t N, int64_t F, int64_t K, int likelihood, double* grad_pos, double* grad_neg) {
    return nmf_grad_x_api<double>(h, Y, mask, X, D, N, F, K, likelihood, grad_pos, grad_neg);
}
int dcp_nmf_gauss_logp_f32(dcp_handle* h, const float* Y, const float* mask, const float* X, const float* D,
                           int64_t N, int64_t F, int64_t K, double scale, double* out) {
    return nmf_gauss_logp_api<float>(h, Y, mask, X, D, N, F, K, scale, out);
}
int dcp_nmf_gauss_logp_f64(dcp_handle* h, const double* Y, const double* mask, const double* X, const double* D,
                           int64_t N, int64_t F, int64_t K, double scale, double* out) {
    return nmf_gauss_logp_api<double>(h, Y, mask, X, D, N, F, K, scale, out);
}
int dcp_nmf_apply_f32(dcp_handle* h, const float* D, const float* P, const float* Q, double alpha,
                      float* D_new, int64_t K, int64_t F, double* maxdiff) {
    return nmf_apply_api<float>(h, D, P, Q, alpha, D_new, K, F, maxdiff);
}
int dcp_nmf_apply_f64(dcp_handle* h, const double* D, const double* P, const double* Q, double alpha,
                      double* D_new, int64_t K, int64_t F, double* maxdiff) {
    return nmf_apply_api<double>(h, D, P, Q, alpha, D_new, K, F, maxdiff);
}
int dcp_axpby_f32(dcp_handle* h, int64_t n, double a, const float* x, double b, float* y) {
    return axpby_api<float>(h, n, a, x, b, y);
}
int dcp_axpby_f64(dcp_handle* h, int64_t n, double a, const double* x, double b, double* y) {
    return axpby_api<double>(h, n, a, x, b, y);
}

int64_t dcp_nmf_mu_stats_width(int64_t F, int64_t K, int likelihood, int masked) {
    return nmf_stats_width(F, K, likelihood, masked != 0);
}

int dcp_nmf_mu_f32(dcp_handle* h, const float* Y, const float* mask, float* X, float* D, int64_t N,
                   int64_t F, int64_t K, int likelihood, float tol, int maxiter, int* it_out,
                   float* last_maxdiff, float* resid_trace) {
    return nmf_mu_solve<float>(h, Y, mask, X, D, N, F, K, likelihood, tol, maxiter, it_out,
                               last_maxdiff, resid_trace);
}
int dcp_nmf_mu_f64(dcp_handle* h, const double* Y, const double* mask, double* X, double* D,
                   int64_t N, int64_t F, int64_t K, int likelihood, double tol, int maxiter,
                   int* it_out, double* last_maxdiff, double* resid_trace) {
    return nmf_mu_solve<double>(h, Y, mask, X, D, N, F, K, likelihood, tol, maxiter, it_out,
                                last_maxdiff, resid_trace);
}
int dcp_nmf_mu_sharded_f32(dcp_handle* h, const float* Y, const float* mask, float* X, float* D, int64_t N,
                           int64_t F, int64_t K, int likelihood, float tol, int maxiter, int* it_out,
                           float* last_maxdiff) {
    return nmf_mu_solve<float>(h, Y, mask, X, D, N, F, K, likelihood, tol, maxiter, it_out,
                               last_maxdiff, nullptr, true);
}
int dcp_nmf_mu_sharded_f64(dcp_handle* h, const double* Y, const double* mask, double* X, double* D,
                           int64_t N, int64_t F, int64_t K, int likelihood, double tol, int maxiter,
                           int* it_out, double* last_maxdiff) {
    return nmf_mu_solve<double>(h, Y, mask, X, D, N, F, K, likelihood, tol, maxiter, it_out,
                                last_maxdiff, nullptr, true);
}
int dcp_nmf_mu_stats_f32(dcp_handle* h, const float* Y, const float* mask, const float* X,
                         float* X_out, const float* D, int64_t N, int64_t F, int64_t K,
                         int likelihood, float* stats) {
    return nmf_mu_stats_api<float>(h, Y, mask, X, X_out, D, N, F, K, likelihood, stats);
}
int dcp_nmf_mu_stats_f64(dcp_handle* h, const double* Y, const double* mask, const double* X,
                         double* X_out, const double* D, int64_t N, int64_t F, int64_t K,
                         int likelihood, double* stats) {
    return nmf_mu_stats_api<double>(h, Y, mask, X, X_out, D, N, F, K, likelihood, stats);
}
int dcp_nmf_mask_prepare_f32(dcp_handle* h, const float* Y, const float* mask, int64_t N, int64_t F,
                             float* Ym, uint32_t* bits, int* binary) {
    return nmf_mask_prepare_api<float>(h, Y, mask, N, F, Ym, bits, binary);
}
int dcp_nmf_mask_prepare_f64(dcp_handle* h, const double* Y, const double* mask, int64_t N, int64_t F,
                             double* Ym, uint32_t* bits, int* binary) {
    return nmf_mask_prepare_api<double>(h, Y, mask, N, F, Ym, bits, binary);
}
int64_t dcp_nmf_mask_bits_words(int64_t N, int64_t F) { return (int64_t)mask_bits_words(N, F); }
int dcp_nmf_mu_stats_prepared_f32(dcp_handle* h, const float* Ym, const float* mask, const uint32_t* bits,
                                  const float* X, float* X_out, const float* D, int64_t N, int64_t F,
                                  int64_t K, int likelihood, float* stats) {
    return nmf_mu_stats_prepared_api<float>(h, Ym, mask, bits, X, X_out, D, N, F, K, likelihood, stats);
}
int dcp_nmf_mu_stats_prepared_f64(dcp_handle* h, const double* Ym, const double* mask, const uint32_t* bits,
                                  const double* X, double* X_out, const double* D, int64_t N, int64_t F,
                                  int64_t K, int likelihood, double* stats) {
    return nmf_mu_stats_prepared_api<double>(h, Ym, mask, bits, X, X_out, D, N, F, K, likelihood, stats);
}
int dcp_nmf_mu_update_f32(dcp_handle* h, const float* stats, const float* D, float* D_new,
                          int64_t F, int64_t K, int likelihood, int masked, float* maxdiff_dev,
                          float* maxdiff_next) {
    return nmf_mu_update_api<float>(h, stats, D, D_new, F, K, likelihood, masked, maxdiff_dev,
                                    maxdiff_next);
}
int dcp_nmf_mu_update_f64(dcp_handle* h, const double* stats, const double* D, double* D_new,
                          int64_t F, int64_t K, int likelihood, int masked, double* maxdiff_dev,
                          double* maxdiff_next) {
    return nmf_mu_update_api<double>(h, stats, D, D_new, F, K, likelihood, masked, maxdiff_dev,
                                     maxdiff_next);
}
int dcp_nmf_residual_f32(dcp_handle* h, const float* Y, const float* mask, const float* X,
                         const float* D, int64_t N, int64_t F, int64_t K, double* out) {
    return nmf_residual_api<float>(h, Y, mask, X, D, N, F, K, out);
}
int dcp_nmf_residual_f64(dcp_handle* h, const double* Y, const double* mask, const double* X,
                         const double* D, int64_t N, int64_t F, int64_t K, double* out) {
    return nmf_residual_api<double>(h, Y, mask, X, D, N, F, K, out);
}

}  // extern "C"
