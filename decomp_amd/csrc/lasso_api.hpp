// C-ABI glue for the LASSO solvers: argument checks, workspace planning, prox selection.
#pragma once
#include "lasso_impl.hpp"

namespace dcp {

template <class T>
inline int lasso_api(dcp_handle* h, const T* Y, const real_t<T>* mask, int mask_ndim, const T* A, T* X,
                     int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter, int method,
                     int positive, int* it_out, const LassoExtra& extra = LassoExtra()) {
    typedef real_t<T> R;
    if (!h) return DCP_ERR_INVALID;
    if (!Y || !A || !X || !it_out) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (N <= 0 || F <= 0 || K <= 0) return fail(h, DCP_ERR_INVALID, "sizes must be positive");
    if (N > 0x7fffffffLL || F > 0x7fffffffLL || K > 0x7fffffffLL)
        return fail(h, DCP_ERR_INVALID, "dimension exceeds 2^31-1");
    if (mask_ndim < 0 || mask_ndim > 2 || (mask_ndim != 0 && !mask) || (mask_ndim == 0 && mask))
        return fail(h, DCP_ERR_INVALID, "mask / mask_ndim mismatch");
    if (method < DCP_LASSO_ISTA || method > DCP_LASSO_ADMM) return fail(h, DCP_ERR_INVALID, "bad method");
    if (method == DCP_LASSO_PARALLEL_CD && (!extra.order || extra.order_rows <= 0))
        return fail(h, DCP_ERR_INVALID, "parallel_cd needs the shuffle table: call dcp_lasso_pcd_*");
    if (method == DCP_LASSO_ADMM && !(extra.rho > 0.0)) return fail(h, DCP_ERR_INVALID, "admm: rho must be > 0");
    if (positive && scalar_traits<T>::is_complex)
        return fail(h, DCP_ERR_INVALID, "positive solvers need a real dtype (lasso.py:92)");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    WsPlan plan;
    lasso_plan<T>(plan, N, F, K, mask_ndim, method);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    LassoWs<T> w;
    DCP_TRY(lasso_carve<T>(h, w, N, F, K, mask_ndim, method));
    if constexpr (scalar_traits<T>::is_complex) {
        return lasso_solve<T, PROX_COMPLEX>(h, Y, mask, mask_ndim, A, X, N, F, K, (R)alpha, (R)tol,
                                            maxiter, method, it_out, w, extra);
    } else {
        if (positive)
            return lasso_solve<T, PROX_POSITIVE>(h, Y, mask, mask_ndim, A, X, N, F, K, (R)alpha, (R)tol,
                                                 maxiter, method, it_out, w, extra);
        return lasso_solve<T, PROX_REAL>(h, Y, mask, mask_ndim, A, X, N, F, K, (R)alpha, (R)tol,
                                         maxiter, method, it_out, w, extra);
    }
}

}  // namespace dcp
