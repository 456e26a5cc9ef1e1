// Kernels of the two remaining LASSO solvers of the reference:
//   decomp/lasso.py:448-523   parallel ("shotgun") coordinate descent, plain and 2-D mask
//   decomp/lasso.py:586-657   ADMM, plain and 2-D mask (one K x K system per row)
//   decomp/math_utils/linalg.py:9-16  batched matrix inverse
//
// Both keep the structure of the proximal-gradient solvers: the per-iteration product
// (x.AAt, or V.(AAt + rho I)^-1) is a GEMM and everything elementwise around it lives in
// that GEMM's epilogue, so an iteration is one launch.
#pragma once
#include "handle.hpp"
#include "kernels_small.hpp"
#include "scalar.hpp"

namespace dcp {

template <int PROX, class T>
__device__ __forceinline__ T prox_apply(T z, real_t<T> thr);   // lasso_impl.hpp

// ---- parallel coordinate descent --------------------------------------------------------------
// lasso.py:476-482 fused into the GEMM that yields back = x0.AAt (or ((x0 A) o M) A^H):
//   x_new = S(x0 + 1.0 * (yAt - back), alpha_k [* rowscale_n]);  dx = x_new - x0
//   stop flag |dx| - tol_k >= 0 (check iterations; x_new is stored then, it is what a
//   converged call returns);  x_next = x0 + dx * select_k, select_k = (order_k < p)
template <class T, int PROX>
struct EpiPcdStep {
    typedef real_t<T> R;
    const T* yAt;
    const T* x0;
    T* xnew;            // written on check iterations only
    T* xnext;
    long ld;
    const R* alpha;     // [K]
    const R* tolk;      // [K]
    const R* rowscale;  // nullable [N]
    const int* order;   // [K] this iteration's row of the shuffle table
    int p;
    int check;
    int* flag;
    __device__ __forceinline__ void operator()(int r, int c, T back, int) const {
        const long i = (long)r * ld + c;
        const T x = x0[i];
        const T z = add(x, sub(yAt[i], back));
        R thr = alpha[c];
        if (rowscale != nullptr) thr = alpha[c] * rowscale[r];
        const T xn = prox_apply<PROX>(z, thr);
        const T d = sub(xn, x);
        if (check) {
            xnew[i] = xn;
            if (!((absval(d) - tolk[c]) < R(0))) *flag = 1;
        }
        const R sel = (order[c] < p) ? R(1) : R(0);
        xnext[i] = add(x, scale(d, sel));
    }
};

// ---- ADMM -------------------------------------------------------------------------------------
// lasso.py:607-617 fused into the GEMM x_new = V.Minv, V = yAt + rho (z - u):
//   z = S(x_new + u, alpha_k / rho);  flags |x - x_new| - tol_k >= 0, |z - x_new| - tol_k >= 0
//   x <- x_new;  u <- u + x_new - z;  V_next = yAt + rho (z - u)
// x and u are updated in place (element-local); V is the GEMM operand, hence double buffered.
template <class T, int PROX>
struct EpiAdmmStep {
    typedef real_t<T> R;
    const T* yAt;
    T* x;
    T* u;
    T* vnext;
    long ld;
    const R* alpha;     // [K]
    const R* tolk;      // [K]
    R rho;
    int check;
    int* flag;
    __device__ __forceinline__ void operator()(int r, int c, T xn, int) const {
        const long i = (long)r * ld + c;
        const T uo = u[i];
        const T z = prox_apply<PROX>(add(xn, uo), alpha[c] / rho);
        if (check) {
            const bool ok = ((absval(sub(x[i], xn)) - tolk[c]) < R(0)) &&
                            ((absval(sub(z, xn)) - tolk[c]) < R(0));
            if (!ok) *flag = 1;
        }
        x[i] = xn;
        const T un = sub(add(uo, xn), z);
        u[i] = un;
        vnext[i] = add(yAt[i], scale(sub(z, un), rho));
    }
};

// V0 = yAt + rho * (z - u) with z = u = x   (lasso.py:605-609, first pass)
template <class T>
__global__ void __launch_bounds__(256) admm_init_kernel(const T* __restrict__ yAt, const T* __restrict__ x,
                                                        long n, real_t<T> rho, T* __restrict__ u,
                                                        T* __restrict__ v) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const T xv = x[i];
        u[i] = xv;
        v[i] = add(yAt[i], scale(sub(xv, xv), rho));
    }
}

// ---- matrix inverse (linalg.py:9-16) ------------------------------------------------------------
// The systems are AAt + rho I: Hermitian positive definite, so Gauss-Jordan needs no pivoting.
// Worked in double (complex double) whatever the problem dtype -- the reference's own inverse is
// double as well (its `rho * eye(K)` promotes, see oracle/lasso.py).
template <class T> struct work_of { typedef double type; };
template <class R> struct work_of<cx<R> > { typedef cx<double> type; };
template <class T> using work_t = typename work_of<T>::type;

__device__ __forceinline__ double to_work(float v) { return (double)v; }
__device__ __forceinline__ double to_work(double v) { return v; }
template <class R> __device__ __forceinline__ cx<double> to_work(cx<R> v) {
    return cx<double>{(double)v.re, (double)v.im};
}
template <class T> __device__ __forceinline__ T from_work(work_t<T> v);
template <> __device__ __forceinline__ float from_work<float>(double v) { return (float)v; }
template <> __device__ __forceinline__ double from_work<double>(double v) { return v; }
template <> __device__ __forceinline__ c64 from_work<c64>(cx<double> v) { return c64{(float)v.re, (float)v.im}; }
template <> __device__ __forceinline__ c128 from_work<c128>(cx<double> v) { return v; }

__device__ __forceinline__ double recip(double v) { return 1.0 / v; }
__device__ __forceinline__ cx<double> recip(cx<double> v) {
    const double d = v.re * v.re + v.im * v.im;
    return cx<double>{v.re / d, -v.im / d};
}
__device__ __forceinline__ double add_real(double v, double r) { return v + r; }
__device__ __forceinline__ cx<double> add_real(cx<double> v, double r) { return cx<double>{v.re + r, v.im}; }

// W = (work type) S + rho * I
template <class T>
__global__ void __launch_bounds__(256) inv_load_kernel(const T* __restrict__ S, int K, double rho,
                                                       work_t<T>* __restrict__ W) {
    const long n = (long)K * K;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        work_t<T> v = to_work(S[i]);
        if (i / K == i % K) v = add_real(v, rho);
        W[i] = v;
    }
}
template <class T>
__global__ void __launch_bounds__(256) inv_store_kernel(const work_t<T>* __restrict__ W, long n,
                                                        T* __restrict__ out) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L)
        out[i] = from_work<T>(W[i]);
}

// One Gauss-Jordan elimination step k on a single K x K matrix, out of place (src -> dst) so the
// whole chip can work on it without hazards:
//   dst[k][j] = src[k][j] / piv (j != k), dst[k][k] = 1 / piv
//   dst[i][j] = (j == k ? 0 : src[i][j]) - src[i][k] * dst[k][j]          (i != k)
template <class TW>
__global__ void __launch_bounds__(256) gj_step_kernel(const TW* __restrict__ src, TW* __restrict__ dst,
                                                      int K, int k) {
    const TW ipiv = recip(src[(long)k * K + k]);
    const long n = (long)K * K;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256L) {
        const int i = (int)(e / K), j = (int)(e % K);
        const TW rkj = (j == k) ? ipiv : mul(src[(long)k * K + j], ipiv);
        TW out;
        if (i == k) {
            out = rkj;
        } else {
            const TW base = (j == k) ? zero_of<TW>() : src[e];
            out = sub(base, mul(src[(long)i * K + k], rkj));
        }
        dst[e] = out;
    }
}

// Batched in-place Gauss-Jordan: one workgroup per K x K matrix (the per-row systems of the
// masked ADMM).  Dynamic LDS: 2 K work elements (pivot row and pivot column of the step).
template <class TW>
__global__ void __launch_bounds__(256) gj_batched_kernel(TW* __restrict__ Wm, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gj_lds[];
    TW* rowk = reinterpret_cast<TW*>(gj_lds);
    TW* colk = rowk + K;
    TW* a = Wm + (long)blockIdx.x * K * K;
    const int n = K * K;
    for (int k = 0; k < K; ++k) {
        const TW ipiv = recip(a[(long)k * K + k]);
        for (int j = threadIdx.x; j < K; j += 256) {
            rowk[j] = (j == k) ? ipiv : mul(a[(long)k * K + j], ipiv);
            colk[j] = a[(long)j * K + k];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < n; e += 256) {
            const int i = e / K, j = e % K;
            TW out;
            if (i == k) {
                out = rowk[j];
            } else {
                const TW base = (j == k) ? zero_of<TW>() : a[e];
                out = sub(base, mul(colk[i], rowk[j]));
            }
            a[e] = out;
        }
        __syncthreads();
    }
}

// ---- masked ADMM (lasso.py:621-657) -----------------------------------------------------------
// Per row n:  S_n = (M_n o A) A^H + rho I  (K x K), in the work type.  One workgroup per row.
template <class T>
__global__ void __launch_bounds__(256) admm_mask_system_kernel(const T* __restrict__ A,
                                                               const real_t<T>* __restrict__ mask,
                                                               int K, long F, double rho,
                                                               work_t<T>* __restrict__ Wm) {
    typedef work_t<T> TW;
    const long row = blockIdx.x;
    const real_t<T>* m = mask + row * F;
    TW* out = Wm + row * (long)K * K;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int e = wave; e < K * K; e += 4) {          // one wave per entry, F-long dot product
        const int i = e / K, j = e % K;
        const T* ai = A + (long)i * F;
        const T* aj = A + (long)j * F;
        TW acc = zero_of<TW>();
        for (long f = lane; f < F; f += 64)
            acc = madd(acc, to_work(scale(ai[f], m[f])), to_work(conj_of(aj[f])));
        if constexpr (scalar_traits<T>::is_complex) {
            for (int o = 32; o > 0; o >>= 1) {
                acc.re += __shfl_xor(acc.re, o, 64);
                acc.im += __shfl_xor(acc.im, o, 64);
            }
        } else {
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        }
        if (lane == 0) out[e] = (i == j) ? add_real(acc, rho) : acc;
    }
}

// One ADMM iteration for every row with its own inverse (lasso.py:647-656).  One workgroup per
// row; V is staged in LDS so x, u, V can all be updated in place.
template <class T, int PROX>
__global__ void __launch_bounds__(256) admm_mask_step_kernel(const T* __restrict__ yAt,
                                                             const work_t<T>* __restrict__ Winv,
                                                             T* __restrict__ X, T* __restrict__ U,
                                                             T* __restrict__ V, int K,
                                                             const real_t<T>* __restrict__ alphak,
                                                             const real_t<T>* __restrict__ rowscale,
                                                             const real_t<T>* __restrict__ tolk,
                                                             real_t<T> rho, int check,
                                                             int* __restrict__ flag) {
    typedef real_t<T> R;
    typedef work_t<T> TW;
    extern __shared__ __attribute__((aligned(16))) unsigned char admm_lds[];
    T* vrow = reinterpret_cast<T*>(admm_lds);
    const long row = blockIdx.x;
    const TW* inv = Winv + row * (long)K * K;
    for (int j = threadIdx.x; j < K; j += 256) vrow[j] = V[row * K + j];
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += 256) {
        TW acc = zero_of<TW>();
        for (int j = 0; j < K; ++j) acc = madd(acc, to_work(vrow[j]), inv[(long)j * K + c]);
        const T xn = from_work<T>(acc);
        const long i = row * K + c;
        const T uo = U[i];
        const T z = prox_apply<PROX>(add(xn, uo), (alphak[c] * rowscale[row]) / rho);
        if (check) {
            const bool ok = ((absval(sub(X[i], xn)) - tolk[c]) < R(0)) &&
                            ((absval(sub(z, xn)) - tolk[c]) < R(0));
            if (!ok) *flag = 1;
        }
        X[i] = xn;
        const T un = sub(add(uo, xn), z);
        U[i] = un;
        V[i] = add(yAt[i], scale(sub(z, un), rho));
    }
}

}  // namespace dcp
