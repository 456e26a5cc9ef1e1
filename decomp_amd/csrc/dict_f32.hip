// C ABI: dictionary-learning step, f32 (see include/decomp_hip.h;
// reference decomp/dictionary_learning.py:135-164, decomp/utils/data.py:147-156).
#include "dict_api.hpp"

extern "C" {

int dcp_dict_stats_f32(dcp_handle* h, const float* Y, float* X, const float* D, int64_t Nb, int64_t F, int64_t K,
                        double alpha, int lasso_method, int lasso_iter, double lasso_tol, float* stats,
                        int* lasso_it) {
    return dcp::dict_stats_api<float>(h, (Y), (X), (D), Nb, F, K, alpha, lasso_method, lasso_iter,
                                   lasso_tol, (stats), lasso_it);
}

int dcp_dict_update_f32(dcp_handle* h, const float* stats, double beta, float* A, float* B, const float* D, float* D_new,
                         int64_t F, int64_t K, float* maxdiff_dev) {
    return dcp::dict_update_api<float>(h, (stats), beta, (A), (B), (D), (D_new), F, K, maxdiff_dev);
}

int dcp_dict_step_f32(dcp_handle* h, const float* Y, float* X, const float* D, float* D_new, float* A, float* B, int64_t Nb,
                       int64_t F, int64_t K, double beta, double alpha, int lasso_method, int lasso_iter,
                       double lasso_tol, double* maxdiff, int* lasso_it) {
    return dcp::dict_step_api<float>(h, (Y), (X), (D), (D_new), (A), (B), Nb, F, K, beta, alpha,
                                  lasso_method, lasso_iter, lasso_tol, maxdiff, lasso_it);
}

int dcp_dict_step_async_f32(dcp_handle* h, const float* Y, float* X, const float* D, float* D_new, float* A, float* B, int64_t Nb,
                             int64_t F, int64_t K, double beta, double alpha, int lasso_method, int lasso_iter,
                             double lasso_tol, float* maxdiff_dev, int* lasso_it) {
    return dcp::dict_step_async_api<float>(h, (Y), (X), (D), (D_new), (A), (B), Nb, F, K, beta, alpha,
                                        lasso_method, lasso_iter, lasso_tol, maxdiff_dev, lasso_it);
}

int dcp_gather_rows_f32(dcp_handle* h, const float* in, const int64_t* index, int64_t rows, int64_t cols,
                         float* out) {
    return dcp::gather_rows_api<float>(h, (in), reinterpret_cast<const long long*>(index), rows, cols,
                                    (out));
}

int dcp_dict_mask_step_f32(dcp_handle* h, const float* Y, const float* mask, float* X, const float* D,
                            float* D_new, float* A3, float* B, int64_t Nb, int64_t F, int64_t K, double beta,
                            double alpha, int lasso_method, int lasso_iter, double lasso_tol,
                            double* maxdiff, int* lasso_it) {
    return dcp::dict_mask_step_api<float>(h, (Y), mask, (X), (D), (D_new), (A3), (B), Nb, F,
                                       K, beta, alpha, lasso_method, lasso_iter, lasso_tol, maxdiff,
                                       lasso_it);
}

}  // extern "C"
