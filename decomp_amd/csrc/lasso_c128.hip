// C ABI: dcp_lasso_c128 (see include/decomp_hip.h; reference decomp/lasso.py:97-189).
#include "lasso_api.hpp"

extern "C" int dcp_lasso_c128(dcp_handle* h, const void* Y, const double* mask, int mask_ndim,
                              const void* A, void* X, int64_t N, int64_t F, int64_t K, double alpha,
                              double tol, int maxiter, int method, int positive, int* it_out) {
    return dcp::lasso_api<dcp::c128>(h, reinterpret_cast<const dcp::c128*>(Y), mask, mask_ndim, reinterpret_cast<const dcp::c128*>(A), reinterpret_cast<dcp::c128*>(X), N, F, K, alpha, tol, maxiter, method, positive, it_out);
}
