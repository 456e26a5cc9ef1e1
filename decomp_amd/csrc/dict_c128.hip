// C ABI: dictionary-learning step, c128 (see include/decomp_hip.h;
// reference decomp/dictionary_learning.py:135-164, decomp/utils/data.py:147-156).
#include "dict_api.hpp"

extern "C" {

int dcp_dict_stats_c128(dcp_handle* h, const void* Y, void* X, const void* D, int64_t Nb, int64_t F, int64_t K,
                        double alpha, int lasso_method, int lasso_iter, double lasso_tol, void* stats,
                        int* lasso_it) {
    return dcp::dict_stats_api<dcp::c128>(h, reinterpret_cast<const dcp::c128*>(Y), reinterpret_cast<dcp::c128*>(X), reinterpret_cast<const dcp::c128*>(D), Nb, F, K, alpha, lasso_method, lasso_iter,
                                   lasso_tol, reinterpret_cast<dcp::c128*>(stats), lasso_it);
}

int dcp_dict_update_c128(dcp_handle* h, const void* stats, double beta, void* A, void* B, const void* D, void* D_new,
                         int64_t F, int64_t K, double* maxdiff_dev) {
    return dcp::dict_update_api<dcp::c128>(h, reinterpret_cast<const dcp::c128*>(stats), beta, reinterpret_cast<dcp::c128*>(A), reinterpret_cast<dcp::c128*>(B), reinterpret_cast<const dcp::c128*>(D), reinterpret_cast<dcp::c128*>(D_new), F, K, maxdiff_dev);
}

int dcp_dict_step_c128(dcp_handle* h, const void* Y, void* X, const void* D, void* D_new, void* A, void* B, int64_t Nb,
                       int64_t F, int64_t K, double beta, double alpha, int lasso_method, int lasso_iter,
                       double lasso_tol, double* maxdiff, int* lasso_it) {
    return dcp::dict_step_api<dcp::c128>(h, reinterpret_cast<const dcp::c128*>(Y), reinterpret_cast<dcp::c128*>(X), reinterpret_cast<const dcp::c128*>(D), reinterpret_cast<dcp::c128*>(D_new), reinterpret_cast<dcp::c128*>(A), reinterpret_cast<dcp::c128*>(B), Nb, F, K, beta, alpha,
                                  lasso_method, lasso_iter, lasso_tol, maxdiff, lasso_it);
}

int dcp_dict_step_async_c128(dcp_handle* h, const void* Y, void* X, const void* D, void* D_new, void* A, void* B, int64_t Nb,
                             int64_t F, int64_t K, double beta, double alpha, int lasso_method, int lasso_iter,
                             double lasso_tol, double* maxdiff_dev, int* lasso_it) {
    return dcp::dict_step_async_api<dcp::cx<double>>(h, reinterpret_cast<const dcp::cx<double>*>(Y), reinterpret_cast<dcp::cx<double>*>(X), reinterpret_cast<const dcp::cx<double>*>(D), reinterpret_cast<dcp::cx<double>*>(D_new), reinterpret_cast<dcp::cx<double>*>(A), reinterpret_cast<dcp::cx<double>*>(B), Nb, F, K, beta, alpha,
                                        lasso_method, lasso_iter, lasso_tol, maxdiff_dev, lasso_it);
}

int dcp_gather_rows_c128(dcp_handle* h, const void* in, const int64_t* index, int64_t rows, int64_t cols,
                         void* out) {
    return dcp::gather_rows_api<dcp::c128>(h, reinterpret_cast<const dcp::c128*>(in), reinterpret_cast<const long long*>(index), rows, cols,
                                    reinterpret_cast<dcp::c128*>(out));
}

int dcp_dict_mask_step_c128(dcp_handle* h, const void* Y, const double* mask, void* X, const void* D,
                            void* D_new, void* A3, void* B, int64_t Nb, int64_t F, int64_t K, double beta,
                            double alpha, int lasso_method, int lasso_iter, double lasso_tol,
                            double* maxdiff, int* lasso_it) {
    return dcp::dict_mask_step_api<dcp::c128>(h, reinterpret_cast<const dcp::c128*>(Y), mask, reinterpret_cast<dcp::c128*>(X), reinterpret_cast<const dcp::c128*>(D), reinterpret_cast<dcp::c128*>(D_new), reinterpret_cast<dcp::c128*>(A3), reinterpret_cast<dcp::c128*>(B), Nb, F,
                                       K, beta, alpha, lasso_method, lasso_iter, lasso_tol, maxdiff,
                                       lasso_it);
}

}  // extern "C"
