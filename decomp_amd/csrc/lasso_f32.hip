// C ABI: dcp_lasso_f32 (see include/decomp_hip.h; reference decomp/lasso.py:97-189).
#include "lasso_api.hpp"

extern "C" int dcp_lasso_f32(dcp_handle* h, const float* Y, const float* mask, int mask_ndim,
                              const float* A, float* X, int64_t N, int64_t F, int64_t K, double alpha,
                              double tol, int maxiter, int method, int positive, int* it_out) {
    return dcp::lasso_api<float>(h, Y, mask, mask_ndim, A, X, N, F, K, alpha, tol, maxiter, method, positive, it_out);
}
