// C ABI: the small public helpers of the reference that callers import directly
//   decomp/math_utils/eigen.py:9-20      spectral_radius_Gershgorin (batched)
//   decomp/nmf_methods/grads.py:77-93    Likelihood.update_x / update_d quotient (the rule a
//                                        user-supplied Likelihood inherits)
#include "handle.hpp"
#include "kernels_small.hpp"

using namespace dcp;

namespace {

// One workgroup per matrix of a batch [B, n, n]:  out[b] = max_j sum_i |X[b, i, j]|.
// Column sums run down the rows with the column index on the lanes (coalesced).
template <class T>
__global__ void __launch_bounds__(256) gershgorin_batched_kernel(const T* __restrict__ X, long n,
                                                                 real_t<T>* __restrict__ out) {
    typedef real_t<T> R;
    __shared__ R sh[4];
    const T* a = X + (long)blockIdx.x * n * n;
    R best = 0;
    for (long j = threadIdx.x; j < n; j += 256) {
        R acc = 0;
        for (long i = 0; i < n; ++i) acc += absval(a[i * n + j]);
        best = (acc > best || acc != acc) ? acc : best;
    }
    const R m = block_max_256(best, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = m;
}

template <class T>
int gershgorin_api(dcp_handle* h, const T* X, int64_t batch, int64_t n, real_t<T>* out) {
    if (!h) return DCP_ERR_INVALID;
    if (!X || !out) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (batch < 0 || n <= 0) return fail(h, DCP_ERR_INVALID, "bad size");
    if (batch == 0) return DCP_OK;
    if (batch > 0x7fffffffLL) return fail(h, DCP_ERR_INVALID, "batch exceeds 2^31-1");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL((gershgorin_batched_kernel<T>), dim3((unsigned)batch), dim3(256), 0, h->stream,
                       X, (long)n, out);
    DCP_HIP_OK(h, hipGetLastError());
    return DCP_OK;
}

template <class T>
int mu_quotient_api(dcp_handle* h, const T* cur, const T* pos, const T* neg, int64_t rows, int64_t cols,
                    T* out) {
    if (!h) return DCP_ERR_INVALID;
    if (!cur || !pos || !neg || !out) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (rows < 0 || cols < 0) return fail(h, DCP_ERR_INVALID, "negative size");
    if (rows == 0 || cols == 0) return DCP_OK;
    DCP_HIP_OK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL((mu_quotient_kernel<T>), dim3(grid_for(rows * cols)), dim3(256), 0, h->stream,
                       cur, (long)cols, pos, (long)cols, neg, (long)cols, 0, (long)rows, (long)cols, out,
                       (long)cols);
    DCP_HIP_OK(h, hipGetLastError());
    return DCP_OK;
}

// out = l2_strict(U) (or l2), maxdiff = max |ref - out| on the host: the tail of one MU iteration
// (batch_mu.py:21-22) for a caller that produced U itself (user-supplied Likelihood).
template <class T>
int normalize_diff_api(dcp_handle* h, const T* U, const T* ref, T* out, int64_t K, int64_t F, int strict,
                       double* maxdiff) {
    if (!h) return DCP_ERR_INVALID;
    if (!U || !ref || !out || !maxdiff) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (K <= 0 || F <= 0) return fail(h, DCP_ERR_INVALID, "sizes must be positive");
    if (K > 0x7fffffffLL) return fail(h, DCP_ERR_INVALID, "dimension exceeds 2^31-1");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    WsPlan plan;
    plan.add<T>((size_t)K);
    plan.add<T>(2);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    T* rowmax = ws_alloc<T>(h, (size_t)K);
    T* md = ws_alloc<T>(h, 2);
    if (!rowmax || !md) return fail(h, DCP_ERR_INTERNAL, "workspace plan mismatch");
    hipLaunchKernelGGL((row_normalize_kernel<T>), dim3((unsigned)K), dim3(256), 0, h->stream, U, (long)F,
                       (long)F, strict, ref, (long)F, out, (long)F, rowmax, (T*)nullptr, (T*)nullptr,
                       (T*)nullptr);
    DCP_HIP_OK(h, hipGetLastError());
    hipLaunchKernelGGL((final_max_kernel<T>), dim3(1), dim3(256), 0, h->stream, (const T*)rowmax, (long)K,
                       md);
    DCP_HIP_OK(h, hipGetLastError());
    void* hostv = nullptr;
    DCP_TRY(host_scratch(h, 64, &hostv));
    DCP_HIP_OK(h, hipMemcpyAsync(hostv, md, sizeof(T), hipMemcpyDeviceToHost, h->stream));
    DCP_HIP_OK(h, hipStreamSynchronize(h->stream));
    *maxdiff = (double)(*reinterpret_cast<T*>(hostv));
    return DCP_OK;
}

}  // namespace

extern "C" {

int dcp_l2_normalize_diff_f32(dcp_handle* h, const float* U, const float* ref, float* out, int64_t K,
                              int64_t F, int strict, double* maxdiff) {
    return normalize_diff_api<float>(h, U, ref, out, K, F, strict, maxdiff);
}
int dcp_l2_normalize_diff_f64(dcp_handle* h, const double* U, const double* ref, double* out, int64_t K,
                              int64_t F, int strict, double* maxdiff) {
    return normalize_diff_api<double>(h, U, ref, out, K, F, strict, maxdiff);
}

int dcp_gershgorin_f32(dcp_handle* h, const float* X, int64_t batch, int64_t n, float* out) {
    return gershgorin_api<float>(h, X, batch, n, out);
}
int dcp_gershgorin_f64(dcp_handle* h, const double* X, int64_t batch, int64_t n, double* out) {
    return gershgorin_api<double>(h, X, batch, n, out);
}
int dcp_gershgorin_c64(dcp_handle* h, const void* X, int64_t batch, int64_t n, float* out) {
    return gershgorin_api<c64>(h, reinterpret_cast<const c64*>(X), batch, n, out);
}
int dcp_gershgorin_c128(dcp_handle* h, const void* X, int64_t batch, int64_t n, double* out) {
    return gershgorin_api<c128>(h, reinterpret_cast<const c128*>(X), batch, n, out);
}

int dcp_mu_quotient_f32(dcp_handle* h, const float* cur, const float* pos, const float* neg,
                        int64_t rows, int64_t cols, float* out) {
    return mu_quotient_api<float>(h, cur, pos, neg, rows, cols, out);
}
int dcp_mu_quotient_f64(dcp_handle* h, const double* cur, const double* pos, const double* neg,
                        int64_t rows, int64_t cols, double* out) {
    return mu_quotient_api<double>(h, cur, pos, neg, rows, cols, out);
}

}  // extern "C"
