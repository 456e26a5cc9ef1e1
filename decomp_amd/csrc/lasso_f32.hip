// C ABI: dcp_lasso_f32 (see include/decomp_hip.h; reference decomp/lasso.py:97-189).
#include "lasso_api.hpp"

extern "C" int dcp_lasso_f32(dcp_handle* h, const float* Y, const float* mask, int mask_ndim,
                              const float* A, float* X, int64_t N, int64_t F, int64_t K, double alpha,
                              double tol, int maxiter, int method, int positive, int* it_out) {
    return dcp::lasso_api<float>(h, Y, mask, mask_ndim, A, X, N, F, K, alpha, tol, maxiter, method, positive, it_out);
}

// parallel coordinate descent with the host-supplied shuffle table (lasso.py:448-523)
extern "C" int dcp_lasso_pcd_f32(dcp_handle* h, const float* Y, const float* mask, int mask_ndim,
                                  const float* A, float* X, int64_t N, int64_t F, int64_t K, double alpha,
                                  double tol, int maxiter, int positive, const int32_t* order,
                                  int64_t order_rows, int* it_out) {
    dcp::LassoExtra extra;
    extra.order = order;
    extra.order_rows = order_rows;
    return dcp::lasso_api<float>(h, Y, mask, mask_ndim, A, X, N, F, K, alpha, tol, maxiter,
                               DCP_LASSO_PARALLEL_CD, positive, it_out, extra);
}

// ADMM with an explicit penalty rho (lasso.py:586-657; solve_fastpath passes rho = 1.0)
extern "C" int dcp_lasso_admm_f32(dcp_handle* h, const float* Y, const float* mask, int mask_ndim,
                                   const float* A, float* X, int64_t N, int64_t F, int64_t K, double alpha,
                                   double tol, int maxiter, int positive, double rho, int* it_out) {
    dcp::LassoExtra extra;
    extra.rho = rho;
    return dcp::lasso_api<float>(h, Y, mask, mask_ndim, A, X, N, F, K, alpha, tol, maxiter,
                               DCP_LASSO_ADMM, positive, it_out, extra);
}
