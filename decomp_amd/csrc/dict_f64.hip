// C ABI: dictionary-learning step, f64 (see include/decomp_hip.h;
// reference decomp/dictionary_learning.py:135-164, decomp/utils/data.py:147-156).
#include "dict_api.hpp"

extern "C" {

int dcp_dict_stats_f64(dcp_handle* h, const double* Y, double* X, const double* D, int64_t Nb, int64_t F, int64_t K,
                        double alpha, int lasso_method, int lasso_iter, double lasso_tol, double* stats,
                        int* lasso_it) {
    return dcp::dict_stats_api<double>(h, (Y), (X), (D), Nb, F, K, alpha, lasso_method, lasso_iter,
                                   lasso_tol, (stats), lasso_it);
}

int dcp_dict_update_f64(dcp_handle* h, const double* stats, double beta, double* A, double* B, const double* D, double* D_new,
                         int64_t F, int64_t K, double* maxdiff_dev) {
    return dcp::dict_update_api<double>(h, (stats), beta, (A), (B), (D), (D_new), F, K, maxdiff_dev);
}

int dcp_dict_step_f64(dcp_handle* h, const double* Y, double* X, const double* D, double* D_new, double* A, double* B, int64_t Nb,
                       int64_t F, int64_t K, double beta, double alpha, int lasso_method, int lasso_iter,
                       double lasso_tol, double* maxdiff, int* lasso_it) {
    return dcp::dict_step_api<double>(h, (Y), (X), (D), (D_new), (A), (B), Nb, F, K, beta, alpha,
                                  lasso_method, lasso_iter, lasso_tol, maxdiff, lasso_it);
}

int dcp_dict_step_async_f64(dcp_handle* h, const double* Y, double* X, const double* D, double* D_new, double* A, double* B, int64_t Nb,
                             int64_t F, int64_t K, double beta, double alpha, int lasso_method, int lasso_iter,
                             double lasso_tol, double* maxdiff_dev, int* lasso_it) {
    return dcp::dict_step_async_api<double>(h, (Y), (X), (D), (D_new), (A), (B), Nb, F, K, beta, alpha,
                                        lasso_method, lasso_iter, lasso_tol, maxdiff_dev, lasso_it);
}

int dcp_gather_rows_f64(dcp_handle* h, const double* in, const int64_t* index, int64_t rows, int64_t cols,
                         double* out) {
    return dcp::gather_rows_api<double>(h, (in), reinterpret_cast<const long long*>(index), rows, cols,
                                    (out));
}

int dcp_dict_mask_step_f64(dcp_handle* h, const double* Y, const double* mask, double* X, const double* D,
                            double* D_new, double* A3, double* B, int64_t Nb, int64_t F, int64_t K, double beta,
                            double alpha, int lasso_method, int lasso_iter, double lasso_tol,
                            double* maxdiff, int* lasso_it) {
    return dcp::dict_mask_step_api<double>(h, (Y), mask, (X), (D), (D_new), (A3), (B), Nb, F,
                                       K, beta, alpha, lasso_method, lasso_iter, lasso_tol, maxdiff,
                                       lasso_it);
}

}  // extern "C"
