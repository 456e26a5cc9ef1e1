// C ABI: the small public helpers of the reference that callers import directly
//   decomp/math_utils/eigen.py:9-20      spectral_radius_Gershgorin (batched)
//   decomp/nmf_methods/grads.py:77-93    Likelihood.update_x / update_d quotient (the rule a
//                                        user-supplied Likelihood inherits)
//   decomp/math_utils/linalg.py:9-38     inv (batched np.linalg.inv)
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

// ---- batched inverse (math_utils/linalg.py:9-38) ---------------------------------------------------------
// Gauss-Jordan with partial (row) pivoting on the augmented matrix [A | I] in double precision (complex:
// complex double), one workgroup per matrix.  O(n^3) per matrix on one CU: a utility for the small systems the
// reference inverts (ADMM's (A A^H + rho I)^-1 has its own chip-wide elimination in lasso_extra.hpp).
template <class T> struct inv_work { typedef double type; };
template <class R> struct inv_work<cx<R>> { typedef cx<double> type; };
__device__ __forceinline__ double inv_to_work(float v) { return (double)v; }
__device__ __forceinline__ double inv_to_work(double v) { return v; }
template <class R> __device__ __forceinline__ cx<double> inv_to_work(cx<R> v) { return cx<double>{(double)v.re, (double)v.im}; }
template <class T> __device__ __forceinline__ T inv_from_work(typename inv_work<T>::type v);
template <> __device__ __forceinline__ float inv_from_work<float>(double v) { return (float)v; }
template <> __device__ __forceinline__ double inv_from_work<double>(double v) { return v; }
template <> __device__ __forceinline__ c64 inv_from_work<c64>(cx<double> v) { return c64{(float)v.re, (float)v.im}; }
template <> __device__ __forceinline__ c128 inv_from_work<c128>(cx<double> v) { return v; }
__device__ __forceinline__ double inv_recip(double v) { return 1.0 / v; }
__device__ __forceinline__ cx<double> inv_recip(cx<double> v) {
    const double d = v.re * v.re + v.im * v.im;
    return cx<double>{v.re / d, -v.im / d};
}
__device__ __forceinline__ double inv_mag(double v) { return fabs(v); }
__device__ __forceinline__ double inv_mag(cx<double> v) { return fabs(v.re) + fabs(v.im); }   // as LAPACK's cabs1

template <class T>
__global__ void __launch_bounds__(256) inv_batched_kernel(const T* __restrict__ X, int n,
                                                          typename inv_work<T>::type* __restrict__ Wall,
                                                          typename inv_work<T>::type* __restrict__ colall,
                                                          T* __restrict__ out) {
    typedef typename inv_work<T>::type TW;
    __shared__ double s_best[256];
    __shared__ int s_arg[256];
    const long b = blockIdx.x;
    const int n2 = 2 * n;
    TW* W = Wall + b * (long)n * n2;
    TW* colk = colall + b * (long)n;
    const T* a = X + b * (long)n * n;
    for (long e = threadIdx.x; e < (long)n * n2; e += 256) {
        const int i = (int)(e / n2), j = (int)(e % n2);
        W[e] = (j < n) ? inv_to_work(a[(long)i * n + j]) : ((j - n == i) ? inv_to_work(from_real<T>(1)) : zero_of<TW>());
    }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        // pivot: the row i >= k with the largest |W[i][k]| (ties: the smallest i, as a sequential search)
        double best = -1.0;
        int arg = k;
        for (int i = k + threadIdx.x; i < n; i += 256) {
            const double m = inv_mag(W[(long)i * n2 + k]);
            if (m > best) { best = m; arg = i; }
        }
        s_best[threadIdx.x] = best;
        s_arg[threadIdx.x] = arg;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                const double ob = s_best[threadIdx.x + o];
                const int oa = s_arg[threadIdx.x + o];
                if (ob > s_best[threadIdx.x] || (ob == s_best[threadIdx.x] && oa < s_arg[threadIdx.x])) {
                    s_best[threadIdx.x] = ob;
                    s_arg[threadIdx.x] = oa;
                }
            }
            __syncthreads();
        }
        const int p = s_arg[0];
        __syncthreads();
        if (p != k)
            for (int j = threadIdx.x; j < n2; j += 256) {
                const TW t = W[(long)k * n2 + j];
                W[(long)k * n2 + j] = W[(long)p * n2 + j];
                W[(long)p * n2 + j] = t;
            }
        __syncthreads();
        const TW ipiv = inv_recip(W[(long)k * n2 + k]);
        __syncthreads();
        for (int j = threadIdx.x; j < n2; j += 256) W[(long)k * n2 + j] = mul(W[(long)k * n2 + j], ipiv);
        for (int i = threadIdx.x; i < n; i += 256) colk[i] = W[(long)i * n2 + k];
        __syncthreads();
        // columns < k of the left half are already unit vectors with a zero in row k: skip them
        const int jw = n2 - k;
        for (long e = threadIdx.x; e < (long)n * jw; e += 256) {
            const int i = (int)(e / jw), j = k + (int)(e % jw);
            if (i != k) W[(long)i * n2 + j] = sub(W[(long)i * n2 + j], mul(colk[i], W[(long)k * n2 + j]));
        }
        __syncthreads();
    }
    for (long e = threadIdx.x; e < (long)n * n; e += 256) {
        const int i = (int)(e / n), j = (int)(e % n);
        out[b * (long)n * n + e] = inv_from_work<T>(W[(long)i * n2 + n + j]);
    }
}

template <class T>
int inv_api(dcp_handle* h, const T* X, int64_t batch, int64_t n, T* out) {
    typedef typename inv_work<T>::type TW;
    if (!h) return DCP_ERR_INVALID;
    if (!X || !out) return fail(h, DCP_ERR_INVALID, "null pointer");
    if (batch < 0 || n <= 0 || n > 16384) return fail(h, DCP_ERR_INVALID, "bad size");
    if (batch == 0) return DCP_OK;
    if (batch > 0x7fffffffLL) return fail(h, DCP_ERR_INVALID, "batch exceeds 2^31-1");
    DCP_HIP_OK(h, hipSetDevice(h->device));
    WsPlan plan;
    plan.add<TW>((size_t)batch * n * 2 * n);
    plan.add<TW>((size_t)batch * n);
    DCP_TRY(ws_reserve(h, plan.total));
    ws_reset(h);
    TW* W = ws_alloc<TW>(h, (size_t)batch * n * 2 * n);
    TW* col = ws_alloc<TW>(h, (size_t)batch * n);
    if (!W || !col) return fail(h, DCP_ERR_INTERNAL, "workspace plan mismatch");
    hipLaunchKernelGGL((inv_batched_kernel<T>), dim3((unsigned)batch), dim3(256), 0, h->stream, X, (int)n, W, col,
                       out);
    DCP_HIP_OK(h, hipGetLastError());
    return DCP_OK;
}

}  // namespace

extern "C" {

int dcp_inv_f32(dcp_handle* h, const float* X, int64_t batch, int64_t n, float* out) {
    return inv_api<float>(h, X, batch, n, out);
}
int dcp_inv_f64(dcp_handle* h, const double* X, int64_t batch, int64_t n, double* out) {
    return inv_api<double>(h, X, batch, n, out);
}
int dcp_inv_c64(dcp_handle* h, const void* X, int64_t batch, int64_t n, void* out) {
    return inv_api<c64>(h, reinterpret_cast<const c64*>(X), batch, n, reinterpret_cast<c64*>(out));
}
int dcp_inv_c128(dcp_handle* h, const void* X, int64_t batch, int64_t n, void* out) {
    return inv_api<c128>(h, reinterpret_cast<const c128*>(X), batch, n, reinterpret_cast<c128*>(out));
}


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
