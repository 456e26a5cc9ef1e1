// C ABI: dictionary-learning step, c64 (see include/decomp_hip.h;
// reference decomp/dictionary_learning.py:135-164, decomp/utils/data.py:147-156).
#include "dict_api.hpp"

extern "C" {

int dcp_dict_stats_c64(dcp_handle* h, const void* Y, void* X, const void* D, int64_t Nb, int64_t F, int64_t K,
                        double alpha, int lasso_method, int lasso_iter, double lasso_tol, void* stats,
                        int* lasso_it) {
    return dcp::dict_stats_api<dcp::c64>(h, reinterpret_cast<const dcp::c64*>(Y), reinterpret_cast<dcp::c64*>(X), reinterpret_cast<const dcp::c64*>(D), Nb, F, K, alpha, lasso_method, lasso_iter,
                                   lasso_tol, reinterpret_cast<dcp::c64*>(stats), lasso_it);
}

int dcp_dict_update_c64(dcp_handle* h, const void* stats, double beta, void* A, void* B, const void* D, void* D_new,
                         int64_t F, int64_t K, float* maxdiff_dev) {
    return dcp::dict_update_api<dcp::c64>(h, reinterpret_cast<const dcp::c64*>(stats), beta, reinterpret_cast<dcp::c64*>(A), reinterpret_cast<dcp::c64*>(B), reinterpret_cast<const dcp::c64*>(D), reinterpret_cast<dcp::c64*>(D_new), F, K, maxdiff_dev);
}

int dcp_dict_step_c64(dcp_handle* h, const void* Y, void* X, const void* D, void* D_new, void* A, void* B, int64_t Nb,
                       int64_t F, int64_t K, double beta, double alpha, int lasso_method, int lasso_iter,
                       double lasso_tol, double* maxdiff, int* lasso_it) {
    return dcp::dict_step_api<dcp::c64>(h, reinterpret_cast<const dcp::c64*>(Y), reinterpret_cast<dcp::c64*>(X), reinterpret_cast<const dcp::c64*>(D), reinterpret_cast<dcp::c64*>(D_new), reinterpret_cast<dcp::c64*>(A), reinterpret_cast<dcp::c64*>(B), Nb, F, K, beta, alpha,
                                  lasso_method, lasso_iter, lasso_tol, maxdiff, lasso_it);
}

int dcp_dict_step_async_c64(dcp_handle* h, const void* Y, void* X, const void* D, void* D_new, void* A, void* B, int64_t Nb,
                             int64_t F, int64_t K, double beta, double alpha, int lasso_method, int lasso_iter,
                             double lasso_tol, float* maxdiff_dev, int* lasso_it) {
    return dcp::dict_step_async_api<dcp::cx<float>>(h, reinterpret_cast<const dcp::cx<float>*>(Y), reinterpret_cast<dcp::cx<float>*>(X), reinterpret_cast<const dcp::cx<float>*>(D), reinterpret_cast<dcp::cx<float>*>(D_new), reinterpret_cast<dcp::cx<float>*>(A), reinterpret_cast<dcp::cx<float>*>(B), Nb, F, K, beta, alpha,
                                        lasso_method, lasso_iter, lasso_tol, maxdiff_dev, lasso_it);
}

int dcp_gather_rows_c64(dcp_handle* h, const void* in, const int64_t* index, int64_t rows, int64_t cols,
                         void* out) {
    return dcp::gather_rows_api<dcp::c64>(h, reinterpret_cast<const dcp::c64*>(in), reinterpret_cast<const long long*>(index), rows, cols,
                                    reinterpret_cast<dcp::c64*>(out));
}

int dcp_dict_mask_step_c64(dcp_handle* h, const void* Y, const float* mask, void* X, const void* D,
                            void* D_new, void* A3, void* B, int64_t Nb, int64_t F, int64_t K, double beta,
                            double alpha, int lasso_method, int lasso_iter, double lasso_tol,
                            double* maxdiff, int* lasso_it) {
    return dcp::dict_mask_step_api<dcp::c64>(h, reinterpret_cast<const dcp::c64*>(Y), mask, reinterpret_cast<dcp::c64*>(X), reinterpret_cast<const dcp::c64*>(D), reinterpret_cast<dcp::c64*>(D_new), reinterpret_cast<dcp::c64*>(A3), reinterpret_cast<dcp::c64*>(B), Nb, F,
                                       K, beta, alpha, lasso_method, lasso_iter, lasso_tol, maxdiff,
                                       lasso_it);
}

}  // extern "C"
