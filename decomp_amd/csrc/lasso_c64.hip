// C ABI: dcp_lasso_c64 (see include/decomp_hip.h; reference decomp/lasso.py:97-189).
#include "lasso_api.hpp"

extern "C" int dcp_lasso_c64(dcp_handle* h, const void* Y, const float* mask, int mask_ndim,
                              const void* A, void* X, int64_t N, int64_t F, int64_t K, double alpha,
                              double tol, int maxiter, int method, int positive, int* it_out) {
    return dcp::lasso_api<dcp::c64>(h, reinterpret_cast<const dcp::c64*>(Y), mask, mask_ndim, reinterpret_cast<const dcp::c64*>(A), reinterpret_cast<dcp::c64*>(X), N, F, K, alpha, tol, maxiter, method, positive, it_out);
}
