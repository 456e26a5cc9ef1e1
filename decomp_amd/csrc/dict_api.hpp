// C-ABI glue for the dictionary-learning step.
#pragma once
#include "dict_impl.hpp"
#include "lasso_api.hpp"

namespace dcp {

template <class T>
inline void dict_plan_extra(WsPlan& p, int64_t Nb, int64_t F, int64_t K) {
    typedef real_t<T> R;
    p.add<T>(dict_slab_elems<T>(Nb, F, K));
    p.add<R>((size_t)2 * ((F + 63) / 64) + 512);
    p.add<R>(4);
    const bool pads = dict_pads<T>(F, K);
    const int64_t Kp = pads ? pad64(K) : K, Fp = pads ? pad64(F) : F;
    atom_plan<T>(p, Fp, Kp);
    if (pads) {
        p.add<T>((size_t)Kp * Kp);
        p.add<T>((size_t)Kp * Fp);
        p.add<T>((size_t)Kp * Fp);
    }
}
template <class T>
inline int dict_carve_extra(dcp_handle* h, DictWs<T>& w, int64_t Nb, int64_t F, int64_t K) {
    typedef real_t<T> R;
    w.slab_count = dict_slab_elems<T>(Nb, F, K);
    w.slabs = ws_alloc<T>(h, w.slab_count);
    w.partial = ws_alloc<R>(h, (size_t)2 * ((F + 63) / 64) + 512);
    w.scal = ws_alloc<R>(h, 4);
    const bool pads = dict_pads<T>(F, K);
    const int64_t Kp = pads ? pad64(K) : K, Fp = pads ? pad64(F) : F;
    DCP_TRY(atom_carve<T>(h, w.atom, Fp, Kp));
    if (pads) {
        w.padA = ws_alloc<T>(h, (size_t)Kp * Kp);
        w.padB = ws_alloc<T>(h, (size_t)Kp * Fp);
        w.padD = ws_alloc<T>(h, (size_t)Kp * Fp);
        if (!w.padA || !w.padB || !w.padD) return fail(h, DCP_ERR_INTERNAL, "dict workspace plan");
    }
    if (!w.slabs || !w.partial || !w.scal) return fail(h, DCP_ERR_INTERNAL, "dict workspace plan");
    return DCP_OK;
}

template <class T>
inline int dict_check(dcp_handle* h, const void* a, const void* b, const void* c, int64_t Nb, int64_t F,
                      int64_t K) {
    if (!h) return DCP_ERR_INVALID;
    if (!a || !b || !c) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (Nb <= 0 || F <= 0 || K <= 0) return fail(h, DCP_ERR_INVALID, "sizes must be positive");
    if (Nb > 0x7fffffffLL || F + K > 0x3fffffffLL) return fail(h, DCP_ERR_INVALID, "dimension too large");
    return DCP_OK;
}

// lasso_method arrives as DCP_LASSO_* optionally OR'ed with DCP_LASSO_POSITIVE ('_pos' solvers)
inline bool dict_lasso_method_ok(int lasso_method) {
    const int base = lasso_method & ~DCP_LASSO_POSITIVE;
    return base >= DCP_LASSO_ISTA && base <= DCP_LASSO_ADMM;
}

template <class T>
inline int dict_lasso(dcp_handle* h, const T* Y, const real_t<T>* M, int mask_ndim, const T* D, T* X,
                      int64_t Nb, int64_t F, int64_t K, double alpha, double lasso_tol, int lasso_iter,
                      int lasso_method, int* it, LassoWs<T>& lw) {
    typedef real_t<T> R;
    const int base = lasso_method & ~DCP_LASSO_POSITIVE;
    const bool positive = (lasso_method & DCP_LASSO_POSITIVE) != 0;
    LassoExtra extra;
    extra.no_final_sync = true;            // the statistics product follows on the same stream
    extra.start_prefetch = false;          // the prefetch runs beside the atom sweep (dict_step_core)
    if (base == DCP_LASSO_PARALLEL_CD) {   // the RNG stream of lasso.py:463,481 (dcp_dict_set_pcd_order)
        if (!h->pcd_order || h->pcd_K != K || h->pcd_rows < lasso_iter)
            return fail(h, DCP_ERR_INVALID, "parallel_cd inside the dictionary step: call "
                                            "dcp_dict_set_pcd_order with >= lasso_iter rows of K entries");
        extra.order = h->pcd_order;
        extra.order_rows = h->pcd_rows;
    }
    if constexpr (scalar_traits<T>::is_complex) {
        if (positive) return fail(h, DCP_ERR_INVALID, "positive solvers need a real dtype (lasso.py:92)");
        return lasso_solve<T, PROX_COMPLEX>(h, Y, M, mask_ndim, D, X, Nb, F, K, (R)alpha, (R)lasso_tol,
                                            lasso_iter, base, it, lw, extra);
    } else {
        if (positive)
            return lasso_solve<T, PROX_POSITIVE>(h, Y, M, mask_ndim, D, X, Nb, F, K, (R)alpha,
                                                 (R)lasso_tol, lasso_iter, base, it, lw, extra);
        return lasso_solve<T, PROX_REAL>(h, Y, M, mask_ndim, D, X, Nb, F, K, (R)alpha, (R)lasso_tol,
                                         lasso_iter, base, it, lw, extra);
    }
}

// lasso on this rank's minibatch rows + local x^H [y | x]
template <class T>
inline int dict_stats_core(dcp_handle* h, const T* Y, T* X, const T* D, int64_t Nb, int64_t F, int64_t K,
                           double alpha, int lasso_method, int lasso_iter, double lasso_tol, T* stats,
                           int* lasso_it, LassoWs<T>& lw, DictWs<T>& dw, bool keep_slabs = false,
                           bool settle_later = false) {
    typedef real_t<T> R;
    // (a coordinate-descent solve may leave *it to be settled once its stop flag has landed -- lasso_settle_deferred --
    //  so the iteration count is written straight into the caller's word, or into the handle's when there is none)
    int* it = lasso_it ? lasso_it : &h->lasso_it_sink;
    DCP_TRY(dict_lasso<T>(h, Y, (const R*)nullptr, 0, D, X, Nb, F, K, alpha, lasso_tol, lasso_iter,
                          lasso_method, it, lw));
    DCP_TRY(dict_local_stats<T>(h, Y, X, Nb, F, K, stats, dw, keep_slabs));
    if (!settle_later) DCP_TRY(lasso_settle_deferred(h));
    return DCP_OK;
}

template <class T>
inline int dict_stats_api(dcp_handle* h, const T* Y, T* X, const T* D, int64_t Nb, int64_t F, int64_t K,
                          double alpha, int lasso_method, int lasso_iter, double lasso_tol, T* stats,
                          int* lasso_it) {
    DCP_TRY(dict_check<T>(h, Y, X, D, Nb, F, K));
    if (!stats) return fail(h, DCP_ERR_INVALID, "stats is null");
    if (!dict_lasso_method_ok(lasso_method))
        return fail(h, DCP_ERR_INVALID, "bad lasso method");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    WsPlan plan;
    lasso_plan<T>(plan, Nb, F, K, 0, lasso_method & ~DCP_LASSO_POSITIVE);
    dict_plan_extra<T>(plan, Nb, F, K);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    LassoWs<T> lw;
    DictWs<T> dw;
    DCP_TRY(lasso_carve<T>(h, lw, Nb, F, K, 0, lasso_method & ~DCP_LASSO_POSITIVE));
    DCP_TRY(dict_carve_extra<T>(h, dw, Nb, F, K));
    return dict_stats_core<T>(h, Y, X, D, Nb, F, K, alpha, lasso_method, lasso_iter, lasso_tol, stats,
                              lasso_it, lw, dw);
}

template <class T>
inline int dict_update_api(dcp_handle* h, const T* stats, double beta, T* A, T* B, const T* D, T* Dnew,
                           int64_t F, int64_t K, real_t<T>* maxdiff_dev) {
    typedef real_t<T> R;
    if (!h) return DCP_ERR_INVALID;
    if (!stats || !A || !B || !D || !Dnew || !maxdiff_dev) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (F <= 0 || K <= 0) return fail(h, DCP_ERR_INVALID, "sizes must be positive");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    const bool pads = dict_pads<T>(F, K);
    const int64_t Kp = pads ? pad64(K) : K, Fp = pads ? pad64(F) : F;
    WsPlan plan;
    plan.add<R>((size_t)2 * ((F + 63) / 64) + 512);
    atom_plan<T>(plan, Fp, Kp);
    if (pads) {
        plan.add<T>((size_t)Kp * Kp);
        plan.add<T>((size_t)Kp * Fp);
        plan.add<T>((size_t)Kp * Fp);
    }
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    DictWs<T> dw;
    dw.partial = ws_alloc<R>(h, (size_t)2 * ((F + 63) / 64) + 512);
    DCP_TRY(atom_carve<T>(h, dw.atom, Fp, Kp));
    if (pads) {
        dw.padA = ws_alloc<T>(h, (size_t)Kp * Kp);
        dw.padB = ws_alloc<T>(h, (size_t)Kp * Fp);
        dw.padD = ws_alloc<T>(h, (size_t)Kp * Fp);
        if (!dw.padA || !dw.padB || !dw.padD) return fail(h, DCP_ERR_INTERNAL, "dict workspace plan");
    }
    if (!dw.partial) return fail(h, DCP_ERR_INTERNAL, "dict workspace plan");
    DCP_TRY(dict_update<T>(h, stats, (R)beta, A, B, D, Dnew, F, K, maxdiff_dev, dw));
    if (h->pf_inflight) {     // a registered row prefetch was started beside the sweep: join it
        DCP_TRY(main_after_side(h));
        h->pf_inflight = false;
    }
    return DCP_OK;
}

// dictionary_learning.py:137-164 for one minibatch on one GPU; max|D - D_new| lands in the DEVICE scalar
// maxdiff_dev (caller memory, or the workspace scalar when null), nothing waits for the GPU.
template <class T>
inline int dict_step_core(dcp_handle* h, const T* Y, T* X, const T* D, T* Dnew, T* A, T* B, int64_t Nb,
                          int64_t F, int64_t K, double beta, double alpha, int lasso_method,
                          int lasso_iter, double lasso_tol, real_t<T>* maxdiff_dev, int* lasso_it,
                          real_t<T>** scal_out) {
    typedef real_t<T> R;
    DCP_TRY(dict_check<T>(h, Y, X, D, Nb, F, K));
    if (!Dnew || !A || !B) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (!dict_lasso_method_ok(lasso_method))
        return fail(h, DCP_ERR_INVALID, "bad lasso method");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    WsPlan plan;
    lasso_plan<T>(plan, Nb, F, K, 0, lasso_method & ~DCP_LASSO_POSITIVE);
    dict_plan_extra<T>(plan, Nb, F, K);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    LassoWs<T> lw;
    DictWs<T> dw;
    DCP_TRY(lasso_carve<T>(h, lw, Nb, F, K, 0, lasso_method & ~DCP_LASSO_POSITIVE));
    DCP_TRY(dict_carve_extra<T>(h, dw, Nb, F, K));
    // one GPU: the statistics stay as ordered split-K partials (no [K, F+K] sum is formed: `stats` = null) and
    // are summed by the A / B accumulation itself
    DCP_TRY(dict_stats_core<T>(h, Y, X, D, Nb, F, K, alpha, lasso_method, lasso_iter, lasso_tol, (T*)nullptr,
                               lasso_it, lw, dw, /*keep_slabs=*/true, /*settle_later=*/true));
    // a registered prefetch (dcp_dict_prefetch_rows_bytes: the NEXT minibatch's rows into the OTHER staging block)
    // is started by dict_update in front of the atom sweep and runs on the side stream beside it, where the chip and
    // its HBM are nearly idle (beside the LASSO's iterations it measured 1 % slower, LassoExtra::start_prefetch);
    // joined before the step's last kernel.
    R* md = maxdiff_dev ? maxdiff_dev : dw.scal;
    DCP_TRY(dict_update<T>(h, dw.slabs, (R)beta, A, B, D, Dnew, F, K, md, dw, dw.stat_nslabs));
    if (h->pf_inflight) {
        DCP_TRY(main_after_side(h));
        h->pf_inflight = false;
    }
    DCP_TRY(lasso_settle_deferred(h));   // *lasso_it of a cd solve: its flag landed while the rest was enqueued
    if (scal_out) *scal_out = md;
    return DCP_OK;
}

template <class T>
inline int dict_step_api(dcp_handle* h, const T* Y, T* X, const T* D, T* Dnew, T* A, T* B, int64_t Nb,
                         int64_t F, int64_t K, double beta, double alpha, int lasso_method,
                         int lasso_iter, double lasso_tol, double* maxdiff_host, int* lasso_it) {
    typedef real_t<T> R;
    if (!h) return DCP_ERR_INVALID;
    if (!maxdiff_host) return fail(h, DCP_ERR_INVALID, "null pointer");
    R* md = nullptr;
    DCP_TRY(dict_step_core<T>(h, Y, X, D, Dnew, A, B, Nb, F, K, beta, alpha, lasso_method, lasso_iter, lasso_tol,
                              (R*)nullptr, lasso_it, &md));
    void* hostv = nullptr;
    DCP_TRY(host_scratch(h, 64, &hostv));
    DCP_HIP_OK(h, hipMemcpyAsync(hostv, md, sizeof(R), hipMemcpyDeviceToHost, h->stream));
    DCP_HIP_OK(h, hipStreamSynchronize(h->stream));
    *maxdiff_host = (double)(*reinterpret_cast<R*>(hostv));
    return DCP_OK;
}

// The same step without the host read-back (dictionary_learning.py:161-162 is then evaluated by the caller one
// step late, from the device scalar).
template <class T>
inline int dict_step_async_api(dcp_handle* h, const T* Y, T* X, const T* D, T* Dnew, T* A, T* B, int64_t Nb,
                               int64_t F, int64_t K, double beta, double alpha, int lasso_method,
                               int lasso_iter, double lasso_tol, real_t<T>* maxdiff_dev, int* lasso_it) {
    if (!h) return DCP_ERR_INVALID;
    if (!maxdiff_dev) return fail(h, DCP_ERR_INVALID, "null pointer");
    return dict_step_core<T>(h, Y, X, D, Dnew, A, B, Nb, F, K, beta, alpha, lasso_method, lasso_iter, lasso_tol,
                             maxdiff_dev, lasso_it, nullptr);
}

// dictionary_learning.py:192-225 for one minibatch with a mask (solve_cd_mask).
template <class T>
inline int dict_mask_step_api(dcp_handle* h, const T* Y, const real_t<T>* M, T* X, const T* D, T* Dnew,
                              T* A3, T* B, int64_t Nb, int64_t F, int64_t K, double beta, double alpha,
                              int lasso_method, int lasso_iter, double lasso_tol, double* maxdiff_host,
                              int* lasso_it) {
    typedef real_t<T> R;
    DCP_TRY(dict_check<T>(h, Y, X, D, Nb, F, K));
    if (!M || !Dnew || !A3 || !B || !maxdiff_host) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (!dict_lasso_method_ok(lasso_method))
        return fail(h, DCP_ERR_INVALID, "bad lasso method");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    WsPlan plan;
    lasso_plan<T>(plan, Nb, F, K, 2, lasso_method & ~DCP_LASSO_POSITIVE);
    dict_plan_extra<T>(plan, Nb, F, K);
    plan.add<T>((size_t)Nb * F);   // y o m
    plan.add<T>((size_t)K * F);    // x^H (y o m)
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    LassoWs<T> lw;
    DictWs<T> dw;
    DCP_TRY(lasso_carve<T>(h, lw, Nb, F, K, 2, lasso_method & ~DCP_LASSO_POSITIVE));
    DCP_TRY(dict_carve_extra<T>(h, dw, Nb, F, K));
    T* Ym = ws_alloc<T>(h, (size_t)Nb * F);
    T* sB = ws_alloc<T>(h, (size_t)K * F);
    if (!Ym || !sB) return fail(h, DCP_ERR_INTERNAL, "dict workspace plan");
    int it = 0;
    DCP_TRY(dict_lasso<T>(h, Y, M, 2, D, X, Nb, F, K, alpha, lasso_tol, lasso_iter, lasso_method, &it, lw));
    DCP_TRY(lasso_settle_deferred(h));   // (the masked solvers never defer; kept next to the local it belongs to)
    if (lasso_it) *lasso_it = it;
    // A3 <- beta A3 + x^H (x (x) m)
    hipLaunchKernelGGL((dict_mask_gram_kernel<T>), dim3((unsigned)F), dim3(256), 0, st, (const T*)X, M,
                       (long)Nb, (long)F, (int)K, (R)beta, A3);
    DCP_LAUNCH_OK(h, hipGetLastError());
    // B <- beta B + x^H (y o m)
    hipLaunchKernelGGL((mul_mask_kernel<T>), dim3(grid_for((long)Nb * F)), dim3(256), 0, st, Y, M,
                       (long)Nb, (long)F, (long)F, Ym);
    DCP_LAUNCH_OK(h, hipGetLastError());
    {
        GemmArgs<T> a;
        a.A = X; a.lda = K; a.B = Ym; a.ldb = F; a.M = (int)K; a.N = (int)F; a.K = (int)Nb;
        a.conjA = true;
        plan_splits<FORM_TN>(a, kSplitTarget, kMaxSplits);
        if ((size_t)a.ksplits * K * F > dw.slab_count) return fail(h, DCP_ERR_INTERNAL, "dict slab plan");
        DCP_LAUNCH_OK(h, (gemm<FORM_TN>(st, a, EpiSlab<T>{dw.slabs, (long)F, (long)K * F})));
        hipLaunchKernelGGL((reduce_slabs_kernel<T>), dim3(grid_for((long)K * F)), dim3(256), 0, st,
                           dw.slabs, (long)K * F, a.ksplits, (long)K * F, sB);
        DCP_LAUNCH_OK(h, hipGetLastError());
        hipLaunchKernelGGL((scale_add_kernel<T>), dim3(grid_for((long)K * F)), dim3(256), 0, st,
                           (long)K * F, (R)beta, (const T*)sB, B);
        DCP_LAUNCH_OK(h, hipGetLastError());
    }
    hipLaunchKernelGGL((dict_mask_atom_kernel<T>), dim3((unsigned)K), dim3(256), 0, st, (const T*)A3,
                       (const T*)B, D, (long)F, (int)K, Dnew);
    DCP_LAUNCH_OK(h, hipGetLastError());
    const int mb = grid_for((long)K * F, 256);
    hipLaunchKernelGGL((maxabsdiff_partial_kernel<T>), dim3(mb), dim3(256), 0, st, D, (const T*)Dnew,
                       (long)K * F, dw.partial);
    DCP_LAUNCH_OK(h, hipGetLastError());
    hipLaunchKernelGGL((final_max_kernel<R>), dim3(1), dim3(256), 0, st, (const R*)dw.partial, (long)mb,
                       dw.scal);
    DCP_LAUNCH_OK(h, hipGetLastError());
    void* hostv = nullptr;
    DCP_TRY(host_scratch(h, 64, &hostv));
    DCP_HIP_OK(h, hipMemcpyAsync(hostv, dw.scal, sizeof(R), hipMemcpyDeviceToHost, st));
    DCP_HIP_OK(h, hipStreamSynchronize(st));
    *maxdiff_host = (double)(*reinterpret_cast<R*>(hostv));
    return DCP_OK;
}

template <class T>
inline int gather_rows_api(dcp_handle* h, const T* in, const long long* index, int64_t rows, int64_t cols,
                           T* out) {
    if (!h) return DCP_ERR_INVALID;
    if (!in || !index || !out) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (rows < 0 || cols < 0) return fail(h, DCP_ERR_INVALID, "negative size");
    if (rows == 0 || cols == 0) return DCP_OK;
    DCP_HIP_OK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(grid_for((long)rows * cols, 4096)), dim3(256), 0,
                       h->stream, in, index, (long)rows, (long)cols, out);
    DCP_HIP_OK(h, hipGetLastError());
    return DCP_OK;
}

}  // namespace dcp
