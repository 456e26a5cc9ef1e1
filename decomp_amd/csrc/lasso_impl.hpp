// Batched LASSO / NNLS inner solves on the device (float, double, complex64, complex128).
//
// Reference path restated as kernels:
//   decomp/lasso.py:97-189    solve_fastpath: row-normalise A, rescale alpha / tol / x
//   decomp/lasso.py:192-241   soft thresholds (real, complex, positive)
//   decomp/lasso.py:244-271   one proximal-gradient step, plain and masked
//   decomp/lasso.py:274-445   ista / acc_ista / fista (+ 2-D mask variants)
//   decomp/lasso.py:526-583   coordinate descent (+ mask variant)
//   decomp/math_utils/eigen.py:9-20  Gershgorin bound
//
// Structure on the GPU: the per-iteration x.AAt product is the GEMM; everything the
// reference does elementwise around it (gradient, step, soft threshold, momentum
// extrapolation, |dx| - tol stop test) is the GEMM's epilogue, so one iteration is ONE
// kernel (two when a 2-D mask forces the chained (x A o M) A^H form).  The step size
// 1/L stays on the device; the host only reads a 4-byte flag on every 10th iteration,
// exactly where the reference evaluates its stop test.
// Coordinate descent uses the Gram form (rows of A are unit norm, lasso.py:528):
//   g = y A^H - x AAt;  x_k <- S(g_k + x_k AAt_kk, alpha_k);  g -= dx_k AAt[k, :]
// which is what lasso.py:539-548 computes, at 2NK^2 instead of 2NKF.K flops per sweep.
#pragma once
#include <math.h>

#include "gemm.hpp"
#include "handle.hpp"
#include "kernels_small.hpp"
#include "nmf_impl.hpp"  // DCP_LAUNCH_OK, column_sums
#include "lasso_extra.hpp"  // parallel_cd and admm kernels

namespace dcp {

enum { PROX_REAL = 0, PROX_COMPLEX = 1, PROX_POSITIVE = 2 };

// ---- proximal operators ------------------------------------------------------------------
template <int PROX, class T>
__device__ __forceinline__ T prox_apply(T z, real_t<T> thr) {
    typedef real_t<T> R;
    if constexpr (scalar_traits<T>::is_complex) {
        // lasso.py:223-225: max(|z| - t, 0) * z / (|z| + 1e-15)
        const R r = absval(z);
        const R m = max_np(r - thr, R(0));
        const R den = r + R(1.0e-15);
        T sgn;
        sgn.re = z.re / den;
        sgn.im = z.im / den;
        return scale(sgn, m);
    } else {
        if (PROX == PROX_POSITIVE) {  // lasso.py:241
            const T v = z - thr;
            return max_np(v, T(0));
        }
        // lasso.py:206-207: max(|z| - t, 0) * sign(z)
        const T m = max_np(absval(z) - thr, T(0));
        const T sg = z > T(0) ? T(1) : (z < T(0) ? T(-1) : (z == T(0) ? T(0) : z /*NaN*/));
        return m * sg;
    }
}

// ---- epilogues -------------------------------------------------------------------------------
// One proximal-gradient step fused into the GEMM that yields back = v.AAt (or (vA o M)A^H):
//   z = v + Linv * (yAt - back) ; x_new = S(z, Linv*alpha_k [* rowscale_n]) ;
//   v_next = x_new + coef * (x_new - x_prev) ; violation flag |x_new - x_prev| - tol_k >= 0
template <class T, int PROX>
struct EpiProxStep {
    typedef real_t<T> R;
    const T* yAt;
    const T* v;
    const T* xprev;
    T* xnew;
    T* vnext;  // nullable (plain ista)
    long ld;
    const R* Linv;      // device scalar
    const R* alpha;     // [K]
    const R* tolk;      // [K]
    const R* rowscale;  // nullable [N]  (2-D mask: sum_f mask[n, f])
    R coef;
    int check;
    int* flag;
    __device__ __forceinline__ void operator()(int r, int c, T back, int) const {
        const long i = (long)r * ld + c;
        const R li = Linv[0];
        const T z = add(v[i], scale(sub(yAt[i], back), li));
        R thr = li * alpha[c];
        if (rowscale != nullptr) thr = li * (alpha[c] * rowscale[r]);
        const T xn = prox_apply<PROX>(z, thr);
        xnew[i] = xn;
        const T d = sub(xn, xprev[i]);
        if (vnext != nullptr) vnext[i] = add(xn, scale(d, coef));
        if (check && !((absval(d) - tolk[c]) < R(0))) *flag = 1;
    }
    // float32: the same step on 4 consecutive columns of a row (16-byte loads of y.A^H, v, x_prev and
    // stores of x_new / v_next: the epilogue reads three arrays per element, see epi_vec4)
    static constexpr bool kVec4 = std::is_same<T, float>::value;
    bool vec_ok() const {
        return al16_ptr(yAt) && al16_ptr(v) && al16_ptr(xprev) && al16_ptr(xnew) && al16_ptr(alpha) &&
               al16_ptr(tolk) && (vnext == nullptr || al16_ptr(vnext)) && (ld % 4) == 0;
    }
    __device__ __forceinline__ void vec4(int r, int c0, f32x4 back, int) const {
        if constexpr (std::is_same<T, float>::value) {
            const long i = (long)r * ld + c0;
            const float li = Linv[0];
            const f32x4 y4 = *reinterpret_cast<const f32x4*>(yAt + i);
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(v + i);
            const f32x4 p4 = *reinterpret_cast<const f32x4*>(xprev + i);
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(alpha + c0);
            const float rs = rowscale != nullptr ? rowscale[r] : 1.0f;
            f32x4 xn, vn;
            bool viol = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z = v4[e] + (y4[e] - back[e]) * li;
                float thr = li * a4[e];
                if (rowscale != nullptr) thr = li * (a4[e] * rs);
                xn[e] = prox_apply<PROX>(z, thr);
                const float d = xn[e] - p4[e];
                vn[e] = xn[e] + d * coef;
                if (check && !((fabsf(d) - tolk[c0 + e]) < 0.0f)) viol = true;
            }
            *reinterpret_cast<f32x4*>(xnew + i) = xn;
            if (vnext != nullptr) *reinterpret_cast<f32x4*>(vnext + i) = vn;
            if (viol) *flag = 1;
        }
    }
    // complex64: two consecutive columns of a row (16 bytes) per call, see CplxColEpi
    static constexpr bool kCVec2 = std::is_same<T, c64>::value;
    bool cvec_ok() const {
        return al16_ptr(yAt) && al16_ptr(v) && al16_ptr(xprev) && al16_ptr(xnew) &&
               (vnext == nullptr || al16_ptr(vnext)) && (ld % 2) == 0;
    }
    __device__ __forceinline__ void cvec2(int r, int c0, f32x4 back, int) const {
        if constexpr (std::is_same<T, c64>::value) {
            const long i = (long)r * ld + c0;
            const float li = Linv[0];
            const f32x4 y4 = *reinterpret_cast<const f32x4*>(yAt + i);
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(v + i);
            const f32x4 p4 = *reinterpret_cast<const f32x4*>(xprev + i);
            const float rs = rowscale != nullptr ? rowscale[r] : 1.0f;
            f32x4 xn4, vn4;
            bool viol = false;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const c64 bk{back[2 * e], back[2 * e + 1]};
                const c64 z = add(c64{v4[2 * e], v4[2 * e + 1]}, scale(sub(c64{y4[2 * e], y4[2 * e + 1]}, bk), li));
                float thr = li * alpha[c0 + e];
                if (rowscale != nullptr) thr = li * (alpha[c0 + e] * rs);
                const c64 xn = prox_apply<PROX>(z, thr);
                const c64 d = sub(xn, c64{p4[2 * e], p4[2 * e + 1]});
                const c64 vn = add(xn, scale(d, coef));
                xn4[2 * e] = xn.re; xn4[2 * e + 1] = xn.im;
                vn4[2 * e] = vn.re; vn4[2 * e + 1] = vn.im;
                if (check && !((absval(d) - tolk[c0 + e]) < 0.0f)) viol = true;
            }
            *reinterpret_cast<f32x4*>(xnew + i) = xn4;
            if (vnext != nullptr) *reinterpret_cast<f32x4*>(vnext + i) = vn4;
            if (viol) *flag = 1;
        }
    }
};

// The same step in the "H form" (no 2-D mask):  with H = I - Linv A A^H and r = Linv y A^H, both formed ONCE per
// solve,   v + Linv (y A^H - v A A^H) = v H + r ,   so the epilogue of the v.H product reads ONE [N, K] array (r)
// instead of three (y A^H, v, x_prev) -- x_prev is only touched when the iteration needs x_new - x_prev (momentum,
// or the stop test of iterations 0, 10, ...).  One launch of the dictionary step's 8192 x 512 x 512 iteration is a
// single lock-step round of workgroups (main loop, then epilogue, nothing to overlap them with), so its [N, K]
// epilogue traffic is exposed time: 84 MB -> 34 MB per launch.  Same fixed point, same iterates up to rounding
// (the Gram-form remark of DESIGN.md section 2 applies: a re-association of lasso.py:248-256).
template <class T, int PROX>
struct EpiProxStepH {
    typedef real_t<T> R;
    const T* rs;        // Linv * yAt
    const T* xprev;     // read only when need_prev
    T* xnew;
    T* vnext;           // nullable (plain ista)
    long ld;
    const R* Linv;      // device scalar
    const R* alpha;     // [K]
    const R* tolk;      // [K]
    R coef;
    int check;
    int need_prev;      // check != 0 or vnext != nullptr
    int* flag;
    // last iteration of a solve that returns its latest iterate (ista, fista): x_new / s goes straight to the
    // caller's array (lasso.py:189), one elementwise pass fewer
    T* xfinal = nullptr;
    const R* sdiv = nullptr;
    __device__ __forceinline__ static T unscale(T v, R sk) {
        if constexpr (scalar_traits<T>::is_complex) { v.re = v.re / sk; v.im = v.im / sk; }
        else v = v / sk;
        return v;
    }
    __device__ __forceinline__ void operator()(int r, int c, T back, int) const {
        const long i = (long)r * ld + c;
        const T z = add(back, rs[i]);
        const T xn = prox_apply<PROX>(z, Linv[0] * alpha[c]);
        xnew[i] = xn;
        if (xfinal != nullptr) xfinal[i] = unscale(xn, sdiv[c]);
        if (need_prev) {
            const T d = sub(xn, xprev[i]);
            if (vnext != nullptr) vnext[i] = add(xn, scale(d, coef));
            if (check && !((absval(d) - tolk[c]) < R(0))) *flag = 1;
        }
    }
    static constexpr bool kVec4 = std::is_same<T, float>::value;
    bool vec_ok() const {
        return al16_ptr(rs) && al16_ptr(xprev) && al16_ptr(xnew) && al16_ptr(alpha) && al16_ptr(tolk) &&
               (vnext == nullptr || al16_ptr(vnext)) && (xfinal == nullptr || (al16_ptr(xfinal) && al16_ptr(sdiv))) &&
               (ld % 4) == 0;
    }
    __device__ __forceinline__ void vec4(int r, int c0, f32x4 back, int) const {
        if constexpr (std::is_same<T, float>::value) {
            const long i = (long)r * ld + c0;
            const float li = Linv[0];
            const f32x4 r4 = *reinterpret_cast<const f32x4*>(rs + i);
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(alpha + c0);
            f32x4 xn;
#pragma unroll
            for (int e = 0; e < 4; ++e) xn[e] = prox_apply<PROX>(back[e] + r4[e], li * a4[e]);
            *reinterpret_cast<f32x4*>(xnew + i) = xn;
            if (xfinal != nullptr) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sdiv + c0);
                f32x4 xf;
#pragma unroll
                for (int e = 0; e < 4; ++e) xf[e] = xn[e] / s4[e];
                *reinterpret_cast<f32x4*>(xfinal + i) = xf;
            }
            if (need_prev) {
                const f32x4 p4 = *reinterpret_cast<const f32x4*>(xprev + i);
                f32x4 vn;
                bool viol = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = xn[e] - p4[e];
                    vn[e] = xn[e] + d * coef;
                    if (check && !((fabsf(d) - tolk[c0 + e]) < 0.0f)) viol = true;
                }
                if (vnext != nullptr) *reinterpret_cast<f32x4*>(vnext + i) = vn;
                if (viol) *flag = 1;
            }
        }
    }
    static constexpr bool kCVec2 = std::is_same<T, c64>::value;
    bool cvec_ok() const {
        return al16_ptr(rs) && al16_ptr(xprev) && al16_ptr(xnew) && (vnext == nullptr || al16_ptr(vnext)) &&
               (xfinal == nullptr || al16_ptr(xfinal)) && (ld % 2) == 0;
    }
    __device__ __forceinline__ void cvec2(int r, int c0, f32x4 back, int) const {
        if constexpr (std::is_same<T, c64>::value) {
            const long i = (long)r * ld + c0;
            const float li = Linv[0];
            const f32x4 r4 = *reinterpret_cast<const f32x4*>(rs + i);
            f32x4 xn4;
            c64 xn[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const c64 z{back[2 * e] + r4[2 * e], back[2 * e + 1] + r4[2 * e + 1]};
                xn[e] = prox_apply<PROX>(z, li * alpha[c0 + e]);
                xn4[2 * e] = xn[e].re; xn4[2 * e + 1] = xn[e].im;
            }
            *reinterpret_cast<f32x4*>(xnew + i) = xn4;
            if (xfinal != nullptr) {
                const float s0 = sdiv[c0], s1 = sdiv[c0 + 1];
                *reinterpret_cast<f32x4*>(xfinal + i) = f32x4{xn4[0] / s0, xn4[1] / s0, xn4[2] / s1, xn4[3] / s1};
            }
            if (need_prev) {
                const f32x4 p4 = *reinterpret_cast<const f32x4*>(xprev + i);
                f32x4 vn4;
                bool viol = false;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const c64 d = sub(xn[e], c64{p4[2 * e], p4[2 * e + 1]});
                    const c64 vn = add(xn[e], scale(d, coef));
                    vn4[2 * e] = vn.re; vn4[2 * e + 1] = vn.im;
                    if (check && !((absval(d) - tolk[c0 + e]) < 0.0f)) viol = true;
                }
                if (vnext != nullptr) *reinterpret_cast<f32x4*>(vnext + i) = vn4;
                if (viol) *flag = 1;
            }
        }
    }
};

// AAt <- I - Linv AAt (in place, [K, K]), yAt <- Linv yAt (in place, [N, K]): the H form's two operands; and, in the
// same pass over [N, K], the solver's starting point xs = x * s (lasso.py:131; xsrc null: already done)
template <class T>
__global__ void __launch_bounds__(256) hform_prepare_kernel(T* __restrict__ AAt, long K, T* __restrict__ yAt,
                                                            long nk, const real_t<T>* __restrict__ Linv,
                                                            const T* __restrict__ xsrc,
                                                            const real_t<T>* __restrict__ s, T* __restrict__ xs) {
    typedef real_t<T> R;
    const R li = Linv[0];
    const long kk = K * K;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < kk + nk; i += (long)gridDim.x * 256L) {
        if (i < kk) {
            const long r = i / K, c = i - r * K;
            T h = scale(AAt[i], -li);
            if (r == c) h = add(h, from_real<T>(R(1)));
            AAt[i] = h;
        } else {
            const long j = i - kk;
            yAt[j] = scale(yAt[j], li);
            if (xsrc != nullptr) xs[j] = scale(xsrc[j], s[j % K]);
        }
    }
}

// out = base - acc   (g = yAt - x.AAt for coordinate descent)
template <class T>
struct EpiSubFrom {
    const T* base;
    long ld_base;
    T* out;
    long ld_out;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        out[(long)r * ld_out + c] = sub(base[(long)r * ld_base + c], v);
    }
};

// out = base - acc * mask   (r = y o M - (x A) o M for masked coordinate descent)
template <class T>
struct EpiMaskedResidual {
    const T* base;
    const real_t<T>* mask;
    T* out;
    long ld;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        const long i = (long)r * ld + c;
        out[i] = sub(base[i], scale(v, mask[i]));
    }
};

// ---- small kernels -----------------------------------------------------------------------------
// alpha_k = (alpha / s_k) * nvalid ;  tol_k = tol * s_k        (lasso.py:129-130,135-138)
template <class R>
__global__ void __launch_bounds__(256) lasso_scalars_kernel(const R* __restrict__ s, long K, R alpha,
                                                            R tol, const R* __restrict__ nvalid_dev,
                                                            R nvalid_const, R* __restrict__ alphak,
                                                            R* __restrict__ tolk, int* __restrict__ flag_zero) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *flag_zero = 0;   // the stop flag starts every solve cleared
    const R nv = nvalid_dev != nullptr ? nvalid_dev[0] : nvalid_const;
    for (long k = blockIdx.x * 256L + threadIdx.x; k < K; k += (long)gridDim.x * 256L) {
        alphak[k] = (alpha / s[k]) * nv;
        tolk[k] = tol * s[k];
    }
}

// out[n, k] = x[n, k] * s[k]  (mul != 0)  or  x[n, k] / s[k]     (lasso.py:131,189)
template <class T>
__global__ void __launch_bounds__(256) col_scale_kernel(const T* __restrict__ x, const real_t<T>* __restrict__ s,
                                                        long rows, long K, int mul, T* __restrict__ out) {
    typedef real_t<T> R;
    const long n = rows * K;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const long k = i % K;
        const R sk = s[k];
        T v = x[i];
        if (mul) {
            v = scale(v, sk);
        } else {
            if constexpr (scalar_traits<T>::is_complex) { v.re = v.re / sk; v.im = v.im / sk; }
            else v = v / sk;
        }
        out[i] = v;
    }
}

// Linv = 1 / max_j sum_i |AAt[i, j]|   (eigen.py:20, lasso.py:286), two stages:
// stage 1: partial[s, j] = sum over a stripe of rows of |AAt[i, j]|  (coalesced along j)
template <class T>
__global__ void __launch_bounds__(256) colabs_partial_kernel(const T* __restrict__ a, long K,
                                                             long rows_per_blk,
                                                             real_t<T>* __restrict__ partial) {
    typedef real_t<T> R;
    const long r0 = blockIdx.y * rows_per_blk;
    const long r1 = min(K, r0 + rows_per_blk);
    for (long c = blockIdx.x * 256L + threadIdx.x; c < K; c += (long)gridDim.x * 256L) {
        R acc = 0;
        for (long r = r0; r < r1; ++r) acc += absval(a[r * K + c]);
        partial[blockIdx.y * K + c] = acc;
    }
}
// stage 2 (one workgroup): column sums of the stripes in order, max over columns, inverse.
template <class R>
__global__ void __launch_bounds__(256) gershgorin_finish_kernel(const R* __restrict__ partial, long K,
                                                                int stripes, R* __restrict__ Linv) {
    __shared__ R sh[4];
    R best = 0;
    for (long j = threadIdx.x; j < K; j += 256) {
        // four interleaved running sums (a fixed order): with one, the loop is a chain of
        // dependent L2 round trips (measured 31 us at K = 512 for 128 KiB of partials)
        R a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int s = 0;
        for (; s + 15 < stripes; s += 16) {   // sixteen loads in flight, same four interleaved sums
            R v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partial[(long)(s + u) * K + j];
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                a0 += v[u];
                a1 += v[u + 1];
                a2 += v[u + 2];
                a3 += v[u + 3];
            }
        }
        for (; s + 3 < stripes; s += 4) {
            a0 += partial[(long)s * K + j];
            a1 += partial[(long)(s + 1) * K + j];
            a2 += partial[(long)(s + 2) * K + j];
            a3 += partial[(long)(s + 3) * K + j];
        }
        for (; s < stripes; ++s) a0 += partial[(long)s * K + j];
        const R acc = (a0 + a1) + (a2 + a3);
        best = (acc > best || acc != acc) ? acc : best;
    }
    R m = block_max_256(best, sh);
    if (threadIdx.x == 0) {
        Linv[0] = R(1) / m;
        Linv[2] = m;    // the bound itself (parallel_cd: p = int(K / bound), lasso.py:468)
    }
}

// out[f] = v[f] / count   (mean over the batch of the mask, lasso.py:300-303)
template <class R>
__global__ void __launch_bounds__(256) scale_vec_kernel(const R* __restrict__ v, long n, R divisor,
                                                        R* __restrict__ out) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L)
        out[i] = v[i] / divisor;
}

// sum of a real vector -> out[0]   (one workgroup; sum(mask) for 1-D masks)
template <class R>
__global__ void __launch_bounds__(256) vec_sum_kernel(const R* __restrict__ v, long n, R* __restrict__ out) {
    __shared__ R sh[4];
    R acc = 0;
    for (long i = threadIdx.x; i < n; i += 256) acc += v[i];
    R t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) out[0] = t;
}

// Coordinate-descent sweeps in Gram form.  One wave per row; lane l holds columns l + 64m of the
// row's x and g in registers, plus AAt_kk, alpha_k, tol_k of its columns.
//
// A coordinate whose update is a zero step (x_k unchanged) leaves g untouched, so it commutes
// with everything: per 64-column slot ALL lanes evaluate their candidate step at once, a ballot
// finds the first coordinate (in sweep order) that really moves, only that one is applied
// (g -= dx AAt[k, :], one coalesced row read) and the lanes behind it are re-evaluated.  The
// sequence of applied updates is exactly that of the sequential sweep (lasso.py:539-548); with
// sparse codes a sweep costs ~(#changed coordinates) steps instead of K.
// Sweeps [.., + nsweeps); the FIRST one is a check sweep when check_first != 0 (the reference tests on
// sweeps 0, 10, 20, ...).  `cond` (nullable): the flag of the check sweep launched just before -- zero means its
// test was met and the reference returned there (lasso.py:546-551), so this launch does nothing.
template <class T, int PROX, int MAXM>
__global__ void __launch_bounds__(256) cd_gram_kernel(T* __restrict__ X, T* __restrict__ G,
                                                      const T* __restrict__ AAt,
                                                      const real_t<T>* __restrict__ alphak,
                                                      const real_t<T>* __restrict__ tolk, long rows,
                                                      int K, int nsweeps, int check_first,
                                                      int* __restrict__ flag,
                                                      const int* __restrict__ cond) {
    typedef real_t<T> R;
    // the sweeps behind a check sweep whose test was met are not run (lasso.py:546-551 returns there): decided on
    // the device from the check sweep's flag, so the host does not have to read it before launching them
    if (cond != nullptr && *cond == 0) return;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int M = (K + 63) / 64;
    T x[MAXM], g[MAXM], akk[MAXM];
    R al[MAXM], tl[MAXM];
    // (loads are unconditional on a clamped column and selected afterwards: a load under a lane
    //  predicate becomes its own branch + full wait, i.e. one memory round trip per slot)
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        const int c = lane + 64 * m;
        const bool ok = (m < M && c < K);
        const int cc = ok ? c : 0;
        const T xv = X[row * K + cc], gv = G[row * K + cc], av = AAt[(long)cc * K + cc];
        const R alv = alphak[cc], tlv = tolk[cc];
        x[m] = ok ? xv : zero_of<T>();
        g[m] = ok ? gv : zero_of<T>();
        akk[m] = ok ? av : zero_of<T>();
        al[m] = ok ? alv : R(0);
        tl[m] = ok ? tlv : R(1);
    }
    bool viol = false;
    for (int s = 0; s < nsweeps; ++s) {
        const bool check = check_first && (s == 0);
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            if (m >= M) break;
            const int kend = min(64, K - 64 * m);
            int cursor = 0;   // lanes < cursor of this slot are final for this sweep
            while (true) {
                // candidate step of every lane's coordinate against the current g
                const T z = fmadd(g[m], x[m], akk[m]);
                const T xn = prox_apply<PROX>(z, al[m]);
                const T d = sub(xn, x[m]);
                const bool moves = (lane >= cursor) && (lane < kend) && (abs2(d) != R(0));
                const unsigned long long mask = __ballot(moves);
                if (mask == 0ull) {
                    // every remaining coordinate of the slot is a zero step: |0| - tol_k
                    if (check && lane >= cursor && lane < kend && !((R(0) - tl[m]) < R(0))) viol = true;
                    break;
                }
                const int kk = __ffsll((long long)mask) - 1;     // first moving coordinate
                if (check && lane >= cursor && lane < kk && !((R(0) - tl[m]) < R(0))) viol = true;
                T dk;
                if constexpr (scalar_traits<T>::is_complex) {
                    dk.re = __shfl(d.re, kk, 64);
                    dk.im = __shfl(d.im, kk, 64);
                } else {
                    dk = __shfl(d, kk, 64);
                }
                if (lane == kk) {
                    x[m] = xn;
                    if (check && !((absval(d) - tl[m]) < R(0))) viol = true;
                }
                const T* arow = AAt + (long)(64 * m + kk) * K;
                T ar[MAXM];   // the whole row in flight at once (one L2 round trip per applied step)
#pragma unroll
                for (int mm = 0; mm < MAXM; ++mm) {
                    const int c = lane + 64 * mm;
                    ar[mm] = arow[(mm < M && c < K) ? c : 0];
                }
#pragma unroll
                for (int mm = 0; mm < MAXM; ++mm) {
                    const int c = lane + 64 * mm;
                    const T gn = fmsub(g[mm], dk, ar[mm]);
                    g[mm] = (mm < M && c < K) ? gn : g[mm];
                }
                cursor = kk + 1;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        const int c = lane + 64 * m;
        if (m < M && c < K) {
            X[row * K + c] = x[m];
            G[row * K + c] = g[m];
        }
    }
    if (__ballot(viol) != 0ull && lane == 0) *flag = 1;
}

// The same sweep for dictionaries wider than the register form's 2048 atoms (64 lanes x 32 slots): the
// row's x and g stay in global memory (their own row of X / G: L2-resident, touched by this wave only,
// every lane only ever reads and writes ITS columns c = lane (mod 64), so program order is the only
// ordering needed), one 64-column slot at a time lives in registers while its coordinates are swept, and
// an applied step updates the other slots' g by read-modify-write, eight slots in flight.  Arithmetic
// and update order are those of cd_gram_kernel (identical results where both apply).
template <class T, int PROX>
__global__ void __launch_bounds__(256) cd_gram_wide_kernel(T* __restrict__ X, T* __restrict__ G,
                                                           const T* __restrict__ AAt,
                                                           const real_t<T>* __restrict__ alphak,
                                                           const real_t<T>* __restrict__ tolk, long rows,
                                                           int K, int nsweeps, int check_first,
                                                           int* __restrict__ flag,
                                                           const int* __restrict__ cond) {
    typedef real_t<T> R;
    if (cond != nullptr && *cond == 0) return;   // (see cd_gram_kernel)
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int M = (K + 63) / 64;
    T* xrow = X + row * K;
    T* grow = G + row * K;
    bool viol = false;
    for (int s = 0; s < nsweeps; ++s) {
        const bool check = check_first && (s == 0);
        for (int m = 0; m < M; ++m) {
            const int c0 = lane + 64 * m;
            const bool ok = c0 < K;
            const int cc = ok ? c0 : 0;
            const T xv = xrow[cc], gv = grow[cc], av = AAt[(long)cc * K + cc];
            const R alv = alphak[cc], tlv = tolk[cc];
            T x = ok ? xv : zero_of<T>();
            T g = ok ? gv : zero_of<T>();
            const T akk = ok ? av : zero_of<T>();
            const R al = ok ? alv : R(0);
            const R tl = ok ? tlv : R(1);
            const int kend = min(64, K - 64 * m);
            int cursor = 0;
            while (true) {
                const T z = fmadd(g, x, akk);
                const T xn = prox_apply<PROX>(z, al);
                const T d = sub(xn, x);
                const bool moves = (lane >= cursor) && (lane < kend) && (abs2(d) != R(0));
                const unsigned long long mask = __ballot(moves);
                if (mask == 0ull) {
                    if (check && lane >= cursor && lane < kend && !((R(0) - tl) < R(0))) viol = true;
                    break;
                }
                const int kk = __ffsll((long long)mask) - 1;
                if (check && lane >= cursor && lane < kk && !((R(0) - tl) < R(0))) viol = true;
                T dk;
                if constexpr (scalar_traits<T>::is_complex) {
                    dk.re = __shfl(d.re, kk, 64);
                    dk.im = __shfl(d.im, kk, 64);
                } else {
                    dk = __shfl(d, kk, 64);
                }
                if (lane == kk) {
                    x = xn;
                    if (check && !((absval(d) - tl) < R(0))) viol = true;
                }
                const T* arow = AAt + (long)(64 * m + kk) * K;
                g = ok ? fmsub(g, dk, arow[cc]) : g;          // this slot: the register copy
                for (int m0 = 0; m0 < M; m0 += 8) {               // the other slots: in memory
                    T ar[8], gg[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = lane + 64 * (m0 + u);
                        const int ci = (m0 + u < M && c < K) ? c : 0;
                        ar[u] = arow[ci];
                        gg[u] = grow[ci];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = lane + 64 * (m0 + u);
                        if (m0 + u < M && m0 + u != m && c < K) grow[c] = fmsub(gg[u], dk, ar[u]);
                    }
                }
                cursor = kk + 1;
            }
            if (ok) {
                xrow[c0] = x;
                grow[c0] = g;
            }
        }
    }
    if (__ballot(viol) != 0ull && lane == 0) *flag = 1;
}

// Masked coordinate descent, as written in lasso.py:555-583: with r = (y - x.A) o M kept in
// memory,  x_k <- S(r . conj(A_k) + x_k (A_k . conj(A_k)), alpha_nk);  r -= dx (A_k o M).
// One workgroup per row; F-length dot products by block reduction.  (Parity path for
// lasso.solve(mask=...); not on a BASELINE config.)
template <class T, int PROX>
__global__ void __launch_bounds__(256) cd_mask_kernel(T* __restrict__ X, T* __restrict__ Rres,
                                                      const T* __restrict__ A,
                                                      const real_t<T>* __restrict__ mask,
                                                      const real_t<T>* __restrict__ akk,
                                                      const real_t<T>* __restrict__ alphak,
                                                      const real_t<T>* __restrict__ rowscale,
                                                      const real_t<T>* __restrict__ tolk, int K, long F,
                                                      int nsweeps, int check_last,
                                                      int* __restrict__ flag) {
    typedef real_t<T> R;
    __shared__ R sh_re[4], sh_im[4];
    __shared__ T s_d;
    const long row = blockIdx.x;
    T* r = Rres + row * F;
    const R* mrow = mask + row * F;
    bool viol = false;
    for (int s = 0; s < nsweeps; ++s) {
        const bool check = check_last && (s == nsweeps - 1);
        for (int k = 0; k < K; ++k) {
            const T* ak = A + (long)k * F;
            R pre = 0, pim = 0;
            for (long f = threadIdx.x; f < F; f += 256) {
                const T t = mul(r[f], conj_of(ak[f]));
                pre += real_part(t);
                if constexpr (scalar_traits<T>::is_complex) pim += t.im;
            }
            const R tre = block_sum_256(pre, sh_re);
            R tim = 0;
            if constexpr (scalar_traits<T>::is_complex) tim = block_sum_256(pim, sh_im);
            if (threadIdx.x == 0) {
                T dot;
                if constexpr (scalar_traits<T>::is_complex) { dot.re = tre; dot.im = tim; }
                else dot = tre;
                const T xk = X[row * K + k];
                const T z = add(dot, scale(xk, akk[k]));
                const T xn = prox_apply<PROX>(z, alphak[k] * rowscale[row]);
                const T d = sub(xn, xk);
                if (check && !((absval(d) - tolk[k]) < R(0))) *flag = 1;
                X[row * K + k] = xn;
                s_d = d;
            }
            __syncthreads();
            const T d = s_d;
            if (abs2(d) != R(0))
                for (long f = threadIdx.x; f < F; f += 256)
                    r[f] = sub(r[f], mul(d, scale(ak[f], mrow[f])));
            __syncthreads();
        }
    }
    (void)viol;
}

// akk[k] = sum_f |A[k, f]|^2  is what A_k . conj(A_k) evaluates to; computed by the row-norm
// kernel as s^2.  This helper squares a vector.
template <class R>
__global__ void __launch_bounds__(256) square_vec_kernel(const R* __restrict__ v, long n, R* __restrict__ out) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L)
        out[i] = v[i] * v[i];
}

// Widest dictionary the register form of the coordinate descent takes (64 lanes x 32 slots).  The
// environment variable DCP_CD_REGISTER_LIMIT (tests only) lowers it, so that the memory-resident form
// can be checked bit for bit against the register form on the same problem.
inline int cd_register_limit() {
    const char* e = getenv("DCP_CD_REGISTER_LIMIT");
    int v = e ? atoi(e) : 2048;
    if (v < 1024) v = 1024;     // the launch ladder above 1024 is the only place it is consulted
    if (v > 2048) v = 2048;
    return v;
}

// ---- the solver ----------------------------------------------------------------------------------
template <class T>
struct LassoWs {
    typedef real_t<T> R;
    T* An = nullptr;     // [K,F] row-normalised (and 1-D-masked) design matrix
    T* Ym = nullptr;     // [N,F] y o mask (any mask)
    T* yAt = nullptr;    // [N,K]
    T* AAt = nullptr;    // [K,K]
    T* Am = nullptr;     // [K,F] A o mean(mask)  (2-D mask: Lipschitz bound only)
    T* slabs = nullptr;  // split-K partials of AAt
    T* xb[4] = {nullptr, nullptr, nullptr, nullptr};  // iterates [N,K]
    T* G = nullptr;      // [N,K] CD gradient
    T* T1 = nullptr;     // [N,F] (v A) o M
    R* s = nullptr;      // [K] row norms of A
    R* alphak = nullptr; // [K]
    R* tolk = nullptr;   // [K]
    R* akk = nullptr;    // [K]
    R* rowscale = nullptr;  // [N]
    R* mbar = nullptr;   // [F]
    R* part = nullptr;   // column-sum partials [64, F]
    R* scal = nullptr;   // [4]: Linv, nvalid, L
    work_t<T>* inv_a = nullptr;  // admm: K x K work matrices (ping-pong) or the [N,K,K] per-row systems
    work_t<T>* inv_b = nullptr;
    R* gpart = nullptr;  // [64, K] Gershgorin column-sum stripes
    R* ext1 = nullptr;   // complex: real extended image of a [K,F] operand (4KF reals)
    R* ext2 = nullptr;   // complex: real extended image of AAt (4K^2 reals)
    int* flag = nullptr;
    size_t slab_count = 0;
};

template <class T>
inline void lasso_plan(WsPlan& p, int64_t N, int64_t F, int64_t K, int mask_ndim, int method) {
    typedef real_t<T> R;
    p.add<T>((size_t)K * F);
    if (mask_ndim != 0) p.add<T>((size_t)N * F);
    p.add<T>((size_t)N * K);
    p.add<T>((size_t)K * K);
    if (mask_ndim == 2) p.add<T>((size_t)K * F);
    p.add<T>((size_t)kMaxSplits * K * K);
    for (int i = 0; i < 4; ++i) p.add<T>((size_t)N * K);
    if ((method == DCP_LASSO_CD || method == DCP_LASSO_PARALLEL_CD) && mask_ndim != 2)
        p.add<T>((size_t)N * K);
    if (mask_ndim == 2) p.add<T>((size_t)N * F);
    if (method == DCP_LASSO_ADMM) {
        if (mask_ndim == 2) {
            p.add<work_t<T> >((size_t)N * K * K);
        } else {
            p.add<work_t<T> >((size_t)K * K);
            p.add<work_t<T> >((size_t)K * K);
        }
    }
    for (int i = 0; i < 4; ++i) p.add<R>((size_t)K);
    p.add<R>((size_t)N);
    p.add<R>((size_t)F);
    p.add<R>((size_t)64 * F);
    p.add<R>(4);
    p.add<R>((size_t)64 * K);
    p.add<int>(4);
    if (scalar_traits<T>::is_complex) {
        p.add<R>((size_t)4 * K * F);
        p.add<R>((size_t)4 * K * K);
    }
}

template <class T>
inline int lasso_carve(dcp_handle* h, LassoWs<T>& w, int64_t N, int64_t F, int64_t K, int mask_ndim,
                       int method) {
    typedef real_t<T> R;
    w.An = ws_alloc<T>(h, (size_t)K * F);
    if (mask_ndim != 0) w.Ym = ws_alloc<T>(h, (size_t)N * F);
    w.yAt = ws_alloc<T>(h, (size_t)N * K);
    w.AAt = ws_alloc<T>(h, (size_t)K * K);
    if (mask_ndim == 2) w.Am = ws_alloc<T>(h, (size_t)K * F);
    w.slab_count = (size_t)kMaxSplits * K * K;
    w.slabs = ws_alloc<T>(h, w.slab_count);
    for (int i = 0; i < 4; ++i) w.xb[i] = ws_alloc<T>(h, (size_t)N * K);
    const bool wants_g = (method == DCP_LASSO_CD || method == DCP_LASSO_PARALLEL_CD) && mask_ndim != 2;
    if (wants_g) w.G = ws_alloc<T>(h, (size_t)N * K);
    if (mask_ndim == 2) w.T1 = ws_alloc<T>(h, (size_t)N * F);
    if (method == DCP_LASSO_ADMM) {
        if (mask_ndim == 2) {
            w.inv_a = ws_alloc<work_t<T> >(h, (size_t)N * K * K);
        } else {
            w.inv_a = ws_alloc<work_t<T> >(h, (size_t)K * K);
            w.inv_b = ws_alloc<work_t<T> >(h, (size_t)K * K);
            if (!w.inv_b) return fail(h, DCP_ERR_INTERNAL, "lasso workspace plan mismatch");
        }
        if (!w.inv_a) return fail(h, DCP_ERR_INTERNAL, "lasso workspace plan mismatch");
    }
    w.s = ws_alloc<R>(h, (size_t)K);
    w.alphak = ws_alloc<R>(h, (size_t)K);
    w.tolk = ws_alloc<R>(h, (size_t)K);
    w.akk = ws_alloc<R>(h, (size_t)K);
    w.rowscale = ws_alloc<R>(h, (size_t)N);
    w.mbar = ws_alloc<R>(h, (size_t)F);
    w.part = ws_alloc<R>(h, (size_t)64 * F);
    w.scal = ws_alloc<R>(h, 4);
    w.gpart = ws_alloc<R>(h, (size_t)64 * K);
    w.flag = ws_alloc<int>(h, 4);
    if (scalar_traits<T>::is_complex) {
        w.ext1 = ws_alloc<R>(h, (size_t)4 * K * F);
        w.ext2 = ws_alloc<R>(h, (size_t)4 * K * K);
        if (!w.ext1 || !w.ext2) return fail(h, DCP_ERR_INTERNAL, "lasso workspace plan mismatch");
    }
    if (!w.An || !w.yAt || !w.AAt || !w.slabs || !w.xb[3] || !w.s || !w.alphak || !w.tolk ||
        !w.akk || !w.rowscale || !w.mbar || !w.part || !w.scal || !w.gpart || !w.flag ||
        (mask_ndim != 0 && !w.Ym) || (mask_ndim == 2 && (!w.Am || !w.T1)) ||
        (wants_g && !w.G))
        return fail(h, DCP_ERR_INTERNAL, "lasso workspace plan mismatch");
    return DCP_OK;
}

// C[K,K] = P . Q^H for [K,F] operands, reduction over F split into ordered slabs.
template <class T>
inline int gram_kk(dcp_handle* h, const T* P, const T* Q, int K, int F, LassoWs<T>& w, T* out) {
    GemmArgs<T> g;
    g.A = P; g.lda = F; g.B = Q; g.ldb = F; g.M = K; g.N = K; g.K = F;
    g.conjB = true;
    g.ext_ws = w.ext1;
    // float32, >= 256 atoms: 64 x 64 tiles x 8 splits (tools/gemm_hook_sweep.py 0 512 512 4096: 31 us with the
    // slab sum against 45 us for the 128 x 128 x 64 tile x 32 splits the automatic plan takes)
    if (std::is_same<T, float>::value && K >= 256 && K <= 1024) g.tile = TILE_SMALL;
    plan_splits<FORM_NT>(g, 512, kMaxSplits);
    if ((size_t)g.ksplits * K * K > w.slab_count) return fail(h, DCP_ERR_INTERNAL, "lasso slab plan");
    DCP_LAUNCH_OK(h, (gemm<FORM_NT>(h->stream, g, EpiSlab<T>{w.slabs, K, (long)K * K})));
    hipLaunchKernelGGL((reduce_slabs_kernel<T>), dim3(grid_for((long)K * K)), dim3(256), 0, h->stream,
                       w.slabs, (long)K * K, g.ksplits, (long)K * K, out);
    DCP_LAUNCH_OK(h, hipGetLastError());
    return DCP_OK;
}

// *host = *flag, *flag = 0: the stop flag of a check iteration goes to device-mapped pinned host memory (the host polls
// it one iteration later) and is cleared for the next check.  One 2 us launch instead of a fill kernel in front of the
// iteration, a copy kernel behind it and an event (whose barrier packet idles the GPU ~5 us): 14 -> ~3 us per check.
template <class T = void>
__global__ void flag_publish_kernel(int* __restrict__ flag, int* __restrict__ host) {
    if (threadIdx.x == 0) {
        const int f = *flag;
        *flag = 0;
        *host = f;
    }
}

// Wait for flag_publish_kernel's store: the word was set to the sentinel -1 before that kernel was enqueued.  Polling
// instead of sleeping in a blocking wait, which wakes up late (measured: the GPU sat idle ~2 ms per dictionary step
// once the host ran a step ahead); a stream synchronisation is the fallback after ~2 s.
inline int poll_host_flag(dcp_handle* h, int* host_flag) {
    volatile int* vf = host_flag;
    for (long spin = 0; spin < 400000000L; ++spin) {
        if (*vf != -1) return DCP_OK;
        __builtin_ia32_pause();
    }
    DCP_HIP_OK(h, hipStreamSynchronize(h->stream));
    return DCP_OK;
}

// The dictionary step's deferred *it_out of a coordinate-descent solve (see the Gram-form loop in lasso_solve): called
// once the rest of the step has been enqueued, when the flag has long landed.
inline int lasso_settle_deferred(dcp_handle* h) {
    if (h->lasso_deferred_flag == nullptr) return DCP_OK;
    int* f = h->lasso_deferred_flag;
    h->lasso_deferred_flag = nullptr;
    DCP_TRY(poll_host_flag(h, f));
    if (*f == 0 && h->lasso_deferred_it != nullptr) *h->lasso_deferred_it = h->lasso_deferred_it_met;
    h->lasso_deferred_it = nullptr;
    return DCP_OK;
}

template <class T>
inline int read_flag(dcp_handle* h, int* flag_dev, int* host_flag, bool* violated) {
    DCP_HIP_OK(h, hipMemcpyAsync(host_flag, flag_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    DCP_HIP_OK(h, hipStreamSynchronize(h->stream));
    *violated = (*host_flag != 0);
    return DCP_OK;
}

// scal[0] = 1 / L, scal[2] = L for L = max_j sum_i |M[i, j]|  (eigen.py:20)
template <class T>
inline int gershgorin_bound(dcp_handle* h, const T* M, int K, LassoWs<T>& w) {
    typedef real_t<T> R;
    const int stripes = K >= 64 ? 64 : K;
    const long rows_per = (K + stripes - 1) / stripes;
    hipLaunchKernelGGL((colabs_partial_kernel<T>), dim3(grid_for(K, 64), stripes), dim3(256), 0,
                       h->stream, M, (long)K, rows_per, w.gpart);
    DCP_LAUNCH_OK(h, hipGetLastError());
    hipLaunchKernelGGL((gershgorin_finish_kernel<R>), dim3(1), dim3(256), 0, h->stream,
                       (const R*)w.gpart, (long)K, stripes, w.scal);
    DCP_LAUNCH_OK(h, hipGetLastError());
    return DCP_OK;
}

// Inputs of the two solvers that need more than (alpha, tol, maxiter).
struct LassoExtra {
    const int* order = nullptr;   // parallel_cd: device int32 [order_rows, K] shuffle table
    int64_t order_rows = 0;
    double rho = 1.0;             // admm
    // the caller enqueues more work on the same stream right behind the solve (the dictionary step's x^H [y | x]
    // product): skip the stream synchronisation at the end -- *it_out is known on the host without it
    bool no_final_sync = false;
    // start the dictionary step's registered row prefetch (dcp_dict_prefetch_rows_bytes) right behind the y.A^H
    // product, beside the solver's iterations.  Measured (round 4, configs[2] end to end): 1.720 ms / step against
    // 1.699 when it runs beside the atom sweep instead -- the dictionary step leaves it off.
    bool start_prefetch = false;
};

// solve_fastpath (lasso.py:97-189).  Y [N,F], A [K,F], X [N,K] (in: initial estimate, out:
// solution), mask: null, [F] (mask_ndim 1) or [N,F] (mask_ndim 2).  *it_out as the reference.
template <class T, int PROX>
inline int lasso_solve(dcp_handle* h, const T* Y, const real_t<T>* mask, int mask_ndim, const T* A,
                       T* X, int64_t N64, int64_t F64, int64_t K64, real_t<T> alpha, real_t<T> tol,
                       int maxiter, int method, int* it_out, LassoWs<T>& w,
                       const LassoExtra& extra = LassoExtra()) {
    typedef real_t<T> R;
    hipStream_t st = h->stream;
    h->lasso_deferred_flag = nullptr;   // (a deferral an earlier, failed call never settled dies here)
    h->lasso_deferred_it = nullptr;
    const int N = (int)N64, F = (int)F64, K = (int)K64;
    void* hostv = nullptr;
    DCP_TRY(host_scratch(h, 64, &hostv));
    int* host_flag = reinterpret_cast<int*>(hostv);

    // ---- 1-D mask folds into y and A (lasso.py:120-122); any mask: Ym = y o M ----
    const T* Ause = A;
    const T* Yuse = Y;
    if (mask_ndim == 1) {
        hipLaunchKernelGGL((mul_mask_kernel<T>), dim3(grid_for((long)K * F)), dim3(256), 0, st, A, mask,
                           (long)K, (long)F, 0L, w.An);
        DCP_LAUNCH_OK(h, hipGetLastError());
        Ause = w.An;
        hipLaunchKernelGGL((mul_mask_kernel<T>), dim3(grid_for((long)N * F)), dim3(256), 0, st, Y, mask,
                           (long)N, (long)F, 0L, w.Ym);
        DCP_LAUNCH_OK(h, hipGetLastError());
        Yuse = w.Ym;
        hipLaunchKernelGGL((vec_sum_kernel<R>), dim3(1), dim3(256), 0, st, mask, (long)F, w.scal + 1);
        DCP_LAUNCH_OK(h, hipGetLastError());
    } else if (mask_ndim == 2) {
        hipLaunchKernelGGL((mul_mask_kernel<T>), dim3(grid_for((long)N * F)), dim3(256), 0, st, Y, mask,
                           (long)N, (long)F, (long)F, w.Ym);
        DCP_LAUNCH_OK(h, hipGetLastError());
        Yuse = w.Ym;
        hipLaunchKernelGGL((rowsum_kernel<R>), dim3(N), dim3(256), 0, st, mask, (long)F, (long)F,
                           w.rowscale);
        DCP_LAUNCH_OK(h, hipGetLastError());
    }
    // ---- A scaling (lasso.py:124-131): s = |A_k|, An = A / s, alpha/s, tol*s, x*s ----
    hipLaunchKernelGGL((row_normalize_kernel<T>), dim3(K), dim3(256), 0, st, Ause, (long)F, (long)F, 1,
                       (const T*)nullptr, 0L, w.An, (long)F, (R*)nullptr, w.s);
    DCP_LAUNCH_OK(h, hipGetLastError());
    hipLaunchKernelGGL((lasso_scalars_kernel<R>), dim3(grid_for(K, 64)), dim3(256), 0, st, w.s, (long)K,
                       alpha, tol, mask_ndim == 1 ? w.scal + 1 : (const R*)nullptr,
                       mask_ndim == 2 ? R(1) : R(F), w.alphak, w.tolk, w.flag);
    DCP_LAUNCH_OK(h, hipGetLastError());
    T* xcur = w.xb[0];
    // ista / acc_ista / fista without a 2-D mask: x * s is formed by hform_prepare_kernel in its pass over [N, K]
    const bool hform_family = (method == DCP_LASSO_ISTA || method == DCP_LASSO_ACC_ISTA || method == DCP_LASSO_FISTA) &&
                              mask_ndim != 2;
    if (!hform_family) {
        hipLaunchKernelGGL((col_scale_kernel<T>), dim3(grid_for((long)N * K)), dim3(256), 0, st, (const T*)X,
                           (const R*)w.s, (long)N, (long)K, 1, xcur);
        DCP_LAUNCH_OK(h, hipGetLastError());
    }

    // ---- yAt = (y o M) An^H ----
    {
        GemmArgs<T> a;
        a.A = Yuse; a.lda = F; a.B = w.An; a.ldb = F; a.M = N; a.N = K; a.K = F; a.conjB = true;
        a.ext_ws = w.ext1;
        // deep reduction, few big tiles: ordered split-K partials in the (still unused) iterate buffers
        const long stride = (long)(w.xb[2] - w.xb[1]);
        const int ys = (stride >= (long)N * K && w.xb[3] - w.xb[2] == stride) ? plan_deep_nt(a, 3) : 1;
        if (ys > 1) {
            DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, a, EpiSlab<T>{w.xb[1], K, stride})));
            launch_reduce_slabs<T>(st, (const T*)w.xb[1], stride, ys, (long)N * K, w.yAt);
            DCP_LAUNCH_OK(h, hipGetLastError());
        } else {
            DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, a, EpiStore<T>{w.yAt, K})));
        }
    }
    if (extra.start_prefetch) DCP_TRY(start_registered_prefetch(h));
    const R* rowscale = mask_ndim == 2 ? w.rowscale : nullptr;
    int it = maxiter - 1;
    T* result = xcur;
    bool final_in_place = false;   // the last iteration's epilogue already wrote result / s into X

    // ---- parallel_cd: p = int(K / Gershgorin(A A^H)), unmasked Gram matrix in every variant
    //      (lasso.py:464-470, 503-509); p <= 1 falls back to plain coordinate descent ----
    bool have_gram = false;
    int pcd_p = 0;
    if (method == DCP_LASSO_PARALLEL_CD) {
        DCP_TRY(gram_kk<T>(h, w.An, w.An, K, F, w, w.AAt));
        have_gram = true;
        DCP_TRY(gershgorin_bound<T>(h, w.AAt, K, w));
        R* host_l = reinterpret_cast<R*>(reinterpret_cast<char*>(hostv) + 16);
        DCP_HIP_OK(h, hipMemcpyAsync(host_l, w.scal + 2, sizeof(R), hipMemcpyDeviceToHost, st));
        DCP_HIP_OK(h, hipStreamSynchronize(st));
        const R ratio = (R)K / host_l[0];
        pcd_p = (ratio == ratio && ratio < R(2147483647)) ? (int)ratio : 0;
        if (pcd_p <= 1) {
            if (mask_ndim == 2)
                return fail(h, DCP_ERR_REF_TYPEERROR,
                            "parallel_cd with a full mask and p <= 1: the reference's fallback call "
                            "raises TypeError (lasso.py:509)");
            method = DCP_LASSO_CD;
        } else if (extra.order == nullptr || extra.order_rows < (int64_t)maxiter) {
            return fail(h, DCP_ERR_INVALID, "parallel_cd needs a shuffle table of >= maxiter rows");
        }
    }

    if (method == DCP_LASSO_PARALLEL_CD) {
        // ---------------- parallel coordinate descent (lasso.py:476-484, 515-523) ----------------
        T* X0 = xcur;
        T* Xn = w.xb[1];
        T* Xnew = w.xb[2];
        bool converged = false;
        for (int i = 0; i < maxiter; ++i) {
            const int check = (i % 10 == 0) ? 1 : 0;
            if (check) DCP_HIP_OK(h, hipMemsetAsync(w.flag, 0, sizeof(int), st));
            EpiPcdStep<T, PROX> epi{w.yAt, X0, Xnew, Xn, (long)K, w.alphak, w.tolk, rowscale,
                                    extra.order + (long)i * K, pcd_p, check, w.flag};
            if (mask_ndim == 2) {
                GemmArgs<T> a1;   // T1 = (x0 An) o M
                a1.A = X0; a1.lda = K; a1.B = w.An; a1.ldb = F; a1.M = N; a1.N = F; a1.K = K;
                a1.ext_ws = w.ext1;
                DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a1, EpiMulMask<T>{mask, F, w.T1, F})));
                GemmArgs<T> a2;   // back = T1 An^H
                a2.A = w.T1; a2.lda = F; a2.B = w.An; a2.ldb = F; a2.M = N; a2.N = K; a2.K = F;
                a2.conjB = true;
                a2.ext_ws = w.ext1;
                DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, a2, epi)));
            } else {
                GemmArgs<T> a;    // back = x0 AAt
                a.A = X0; a.lda = K; a.B = w.AAt; a.ldb = K; a.M = N; a.N = K; a.K = K;
                a.ext_ws = w.ext2;
                DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a, epi)));
            }
            if (check) {
                bool viol = true;
                DCP_TRY(read_flag<T>(h, w.flag, host_flag, &viol));
                if (!viol) { it = i; result = Xnew; converged = true; break; }   // lasso.py:479-480
            }
            T* t = X0; X0 = Xn; Xn = t;                                          // x0 += dx * select
        }
        if (!converged) result = X0;
    } else if (method == DCP_LASSO_ADMM && mask_ndim != 2) {
        // ---------------- ADMM (lasso.py:586-618) ----------------
        typedef work_t<T> TW;
        const R rho = (R)extra.rho;
        DCP_TRY(gram_kk<T>(h, w.An, w.An, K, F, w, w.AAt));
        hipLaunchKernelGGL((inv_load_kernel<T>), dim3(grid_for((long)K * K)), dim3(256), 0, st,
                           (const T*)w.AAt, K, extra.rho, w.inv_a);
        DCP_LAUNCH_OK(h, hipGetLastError());
        TW* src = w.inv_a;
        TW* dst = w.inv_b;
        for (int k = 0; k < K; ++k) {
            hipLaunchKernelGGL((gj_step_kernel<TW>), dim3(grid_for((long)K * K)), dim3(256), 0, st,
                               (const TW*)src, dst, K, k);
            TW* t = src; src = dst; dst = t;
        }
        DCP_LAUNCH_OK(h, hipGetLastError());
        T* Minv = w.AAt;     // (AAt + rho I)^-1 replaces AAt
        hipLaunchKernelGGL((inv_store_kernel<T>), dim3(grid_for((long)K * K)), dim3(256), 0, st,
                           (const TW*)src, (long)K * K, Minv);
        DCP_LAUNCH_OK(h, hipGetLastError());
        T* U = w.xb[1];
        T* V = w.xb[2];
        T* Vn = w.xb[3];
        hipLaunchKernelGGL((admm_init_kernel<T>), dim3(grid_for((long)N * K)), dim3(256), 0, st,
                           (const T*)w.yAt, (const T*)xcur, (long)N * K, rho, U, V);
        DCP_LAUNCH_OK(h, hipGetLastError());
        for (int i = 0; i < maxiter; ++i) {
            const int check = (i % 10 == 0) ? 1 : 0;
            if (check) DCP_HIP_OK(h, hipMemsetAsync(w.flag, 0, sizeof(int), st));
            EpiAdmmStep<T, PROX> epi{w.yAt, xcur, U, Vn, (long)K, w.alphak, w.tolk, rho, check, w.flag};
            GemmArgs<T> a;    // x_new = V Minv
            a.A = V; a.lda = K; a.B = Minv; a.ldb = K; a.M = N; a.N = K; a.K = K;
            a.ext_ws = w.ext2;
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a, epi)));
            if (check) {
                bool viol = true;
                DCP_TRY(read_flag<T>(h, w.flag, host_flag, &viol));
                if (!viol) { it = i; break; }                                    // lasso.py:612-614
            }
            T* t = V; V = Vn; Vn = t;
        }
        result = xcur;
    } else if (method == DCP_LASSO_ADMM) {
        // ---------------- ADMM with a 2-D mask: one K x K system per row (lasso.py:621-657) --------
        typedef work_t<T> TW;
        const R rho = (R)extra.rho;
        // dynamic LDS of the two per-row kernels: pivot row + column (2 K work elements) and V (K elements);
        // beyond the default 64 KiB the per-function cap is raised, up to the CU's 160 KiB (K <= 10240 real,
        // 5120 complex -- where ONE row's K x K system is already 0.8 GB; the reference, lasso.py:620-657,
        // holds the same N K^2 array)
        const size_t gj_lds = (size_t)2 * K * sizeof(TW), step_lds = (size_t)K * sizeof(T);
        if (gj_lds > 160 * 1024)
            return fail(h, DCP_ERR_UNSUPPORTED, "masked admm: n_features too large for the per-row inverse");
        if (gj_lds > 65536)
            DCP_HIP_OK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&gj_batched_kernel<TW>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)gj_lds));
        if (step_lds > 65536)
            DCP_HIP_OK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&admm_mask_step_kernel<T, PROX>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds));
        hipLaunchKernelGGL((admm_mask_system_kernel<T>), dim3(N), dim3(256), 0, st, (const T*)w.An, mask,
                           K, (long)F, extra.rho, w.inv_a);
        DCP_LAUNCH_OK(h, hipGetLastError());
        hipLaunchKernelGGL((gj_batched_kernel<TW>), dim3(N), dim3(256), (size_t)2 * K * sizeof(TW), st,
                           w.inv_a, K);
        DCP_LAUNCH_OK(h, hipGetLastError());
        T* U = w.xb[1];
        T* V = w.xb[2];
        hipLaunchKernelGGL((admm_init_kernel<T>), dim3(grid_for((long)N * K)), dim3(256), 0, st,
                           (const T*)w.yAt, (const T*)xcur, (long)N * K, rho, U, V);
        DCP_LAUNCH_OK(h, hipGetLastError());
        for (int i = 0; i < maxiter; ++i) {
            const int check = (i % 10 == 0) ? 1 : 0;
            if (check) DCP_HIP_OK(h, hipMemsetAsync(w.flag, 0, sizeof(int), st));
            hipLaunchKernelGGL((admm_mask_step_kernel<T, PROX>), dim3(N), dim3(256),
                               (size_t)K * sizeof(T), st, (const T*)w.yAt, (const TW*)w.inv_a, xcur, U, V,
                               K, (const R*)w.alphak, (const R*)w.rowscale, (const R*)w.tolk, rho, check,
                               w.flag);
            DCP_LAUNCH_OK(h, hipGetLastError());
            if (check) {
                bool viol = true;
                DCP_TRY(read_flag<T>(h, w.flag, host_flag, &viol));
                if (!viol) { it = i; break; }
            }
        }
        result = xcur;
    } else if (method == DCP_LASSO_CD && mask_ndim != 2) {
        // ---------------- coordinate descent, Gram form ----------------
        if (!have_gram) DCP_TRY(gram_kk<T>(h, w.An, w.An, K, F, w, w.AAt));
        {   // g = yAt - x AAt
            GemmArgs<T> a;
            a.A = xcur; a.lda = K; a.B = w.AAt; a.ldb = K; a.M = N; a.N = K; a.K = K;
            a.ext_ws = w.ext2;
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a, EpiSubFrom<T>{w.yAt, K, w.G, K})));
        }
        // Ten sweeps = a check sweep (0, 10, 20, ...: lasso.py:546-551) launched on its own and the up to nine sweeps
        // behind it in a second launch that the DEVICE skips when the check sweep met the test (its flag is still
        // zero): the codes are then exactly those the reference returns, and nothing on the host has to read the flag
        // between the two launches.  The flag travels to pinned host memory behind them (flag_publish_kernel) and is
        // polled only where the host needs it: to decide on ANOTHER ten sweeps, or for *it_out -- which the
        // dictionary step defers to the end of the step (LassoExtra::no_final_sync), so that its default solve
        // (cd x 10) puts no host round trip into the middle of the step.
        // (Rounds 3-4 ran one launch per ten sweeps with the post-check codes parked in a snapshot and read the flag
        // synchronously: ~37 us of idle GPU per solve in the kernel trace, plus a fill and a copy kernel.)
        int sweep = 0;
        result = xcur;
        const int grid = (N + 3) / 4;
        auto launch = [&](int ns, int check_first, const int* cond) -> int {
#define DCP_CD_LAUNCH(MM)                                                                            \
    hipLaunchKernelGGL((cd_gram_kernel<T, PROX, MM>), dim3(grid), dim3(256), 0, st, xcur, w.G,       \
                       (const T*)w.AAt, (const R*)w.alphak, (const R*)w.tolk, (long)N, K, ns,        \
                       check_first, w.flag, cond)
            // (the register form holds a row's K coefficients in one wave: 64 lanes x up to 32 slots;
            //  wider dictionaries take the memory-resident form -- the reference has no limit, lasso.py:526-552)
            if (K <= 64) DCP_CD_LAUNCH(1);
            else if (K <= 128) DCP_CD_LAUNCH(2);
            else if (K <= 256) DCP_CD_LAUNCH(4);
            else if (K <= 512) DCP_CD_LAUNCH(8);
            else if (K <= 1024) DCP_CD_LAUNCH(16);
            else if (K <= cd_register_limit()) DCP_CD_LAUNCH(32);
            else
                hipLaunchKernelGGL((cd_gram_wide_kernel<T, PROX>), dim3(grid), dim3(256), 0, st, xcur, w.G,
                                   (const T*)w.AAt, (const R*)w.alphak, (const R*)w.tolk, (long)N, K, ns,
                                   check_first, w.flag, cond);
#undef DCP_CD_LAUNCH
            DCP_LAUNCH_OK(h, hipGetLastError());
            return DCP_OK;
        };
        while (sweep < maxiter) {
            int last = sweep + 9;
            if (last > maxiter - 1) last = maxiter - 1;
            // (w.flag is zero here: lasso_scalars_kernel at the start of the solve, flag_publish_kernel afterwards)
            DCP_TRY(launch(1, 1, nullptr));                                     // the check sweep
            if (last > sweep) DCP_TRY(launch(last - sweep, 0, w.flag));         // sweeps sweep + 1 .. last, unless met
            *reinterpret_cast<volatile int*>(host_flag) = -1;                   // sentinel: nothing has landed yet
            hipLaunchKernelGGL(flag_publish_kernel<void>, dim3(1), dim3(64), 0, st, w.flag, host_flag);
            DCP_LAUNCH_OK(h, hipGetLastError());
            if (last == maxiter - 1 && extra.no_final_sync) {
                // the last ten sweeps of a solve inside the dictionary step: *it_out is settled by
                // lasso_settle_deferred() once the rest of the step has been enqueued
                h->lasso_deferred_flag = host_flag;
                h->lasso_deferred_it = it_out;
                h->lasso_deferred_it_met = sweep;
                break;
            }
            DCP_TRY(poll_host_flag(h, host_flag));
            if (*host_flag == 0) {
                it = sweep;
                break;
            }
            sweep = last + 1;
        }
    } else if (method == DCP_LASSO_CD) {
        // ---------------- coordinate descent with a 2-D mask (as written) ----------------
        {   // r = y o M - (x An) o M
            GemmArgs<T> a;
            a.A = xcur; a.lda = K; a.B = w.An; a.ldb = F; a.M = N; a.N = F; a.K = K;
            a.ext_ws = w.ext1;
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a, EpiMaskedResidual<T>{w.Ym, mask, w.T1, (long)F})));
        }
        // A_k . conj(A_k) of the normalised rows, as the reference evaluates it (== 1 up to rounding)
        hipLaunchKernelGGL((row_normalize_kernel<T>), dim3(K), dim3(256), 0, st, (const T*)w.An, (long)F,
                           (long)F, 1, (const T*)nullptr, 0L, w.Am, (long)F, (R*)nullptr, w.akk);
        DCP_LAUNCH_OK(h, hipGetLastError());
        hipLaunchKernelGGL((square_vec_kernel<R>), dim3(grid_for(K, 64)), dim3(256), 0, st,
                           (const R*)w.akk, (long)K, w.akk);
        DCP_LAUNCH_OK(h, hipGetLastError());
        int sweep = 0;
        while (sweep < maxiter) {
            int last = (sweep % 10 == 0) ? sweep : (sweep / 10 + 1) * 10;
            int check_last = 1;
            if (last > maxiter - 1) { last = maxiter - 1; check_last = (last % 10 == 0); }
            const int ns = last - sweep + 1;
            DCP_HIP_OK(h, hipMemsetAsync(w.flag, 0, sizeof(int), st));
            hipLaunchKernelGGL((cd_mask_kernel<T, PROX>), dim3(N), dim3(256), 0, st, xcur, w.T1,
                               (const T*)w.An, mask, (const R*)w.akk, (const R*)w.alphak,
                               (const R*)w.rowscale, (const R*)w.tolk, K, (long)F, ns, check_last,
                               w.flag);
            DCP_LAUNCH_OK(h, hipGetLastError());
            sweep = last + 1;
            if (check_last) {
                bool viol = true;
                DCP_TRY(read_flag<T>(h, w.flag, host_flag, &viol));
                if (!viol) { it = last; break; }
            }
        }
        result = xcur;
    } else {
        // ---------------- ista / acc_ista / fista ----------------
        if (mask_ndim == 2) {
            // Lipschitz bound from (A o mean_batch(M)) A^H   (lasso.py:317)
            DCP_TRY(column_sums<R>(h, mask, F, N, F, w.part, w.mbar));
            hipLaunchKernelGGL((scale_vec_kernel<R>), dim3(grid_for(F, 64)), dim3(256), 0, st,
                               (const R*)w.mbar, (long)F, R(N), w.mbar);
            DCP_LAUNCH_OK(h, hipGetLastError());
            hipLaunchKernelGGL((mul_mask_kernel<T>), dim3(grid_for((long)K * F)), dim3(256), 0, st,
                               (const T*)w.An, (const R*)w.mbar, (long)K, (long)F, 0L, w.Am);
            DCP_LAUNCH_OK(h, hipGetLastError());
            DCP_TRY(gram_kk<T>(h, w.Am, w.An, K, F, w, w.AAt));
        } else {
            DCP_TRY(gram_kk<T>(h, w.An, w.An, K, F, w, w.AAt));
        }
        DCP_TRY(gershgorin_bound<T>(h, w.AAt, K, w));
        const bool hform = (mask_ndim != 2);
        if (hform) {
            hipLaunchKernelGGL((hform_prepare_kernel<T>), dim3(grid_for((long)K * K + (long)N * K)), dim3(256), 0, st,
                               w.AAt, (long)K, w.yAt, (long)N * K, (const R*)w.scal, (const T*)X, (const R*)w.s, xcur);
            DCP_LAUNCH_OK(h, hipGetLastError());
        }

        // Buffer roles (pointers rotate over the four [N,K] buffers):
        //   P = the iterate the stop test compares with (the reference's x0)
        //   V = the point fed to the step (x0 for ista, v / w0 with momentum)
        //   Nw, Vn = this iteration's outputs (x0_new and the next V)
        const bool mom = (method != DCP_LASSO_ISTA);
        T* P = xcur;
        T* V = xcur;
        T* lastP = xcur;
        T* lastNw = xcur;
        double beta = 1.0;
        bool converged = false;
        bool wrote_final = false;
        // The stop test of a check iteration (i % 10 == 0, lasso.py:293) is read ONE iteration late: its flag
        // travels to the host (flag_publish_kernel) while iteration i + 1 is already enqueued, so the GPU does not
        // idle through a host round trip in the middle of every solve (the dictionary step runs ten
        // iterations and checks at i = 0).  Iteration i + 1 only READS iteration i's output, so when the test
        // had passed that output is still intact and i + 1 is simply discarded.
        int pend_i = -1;
        T* pend_x = nullptr;
        auto resolve = [&](bool* stop) -> int {
            *stop = false;
            if (pend_i < 0) return DCP_OK;
            DCP_TRY(poll_host_flag(h, host_flag));
            if (*host_flag == 0) {
                it = pend_i;
                result = pend_x;
                converged = true;
                *stop = true;
            }
            pend_i = -1;
            return DCP_OK;
        };
        for (int i = 0; i < maxiter; ++i) {
            T* Nw = nullptr;
            T* Vn = nullptr;
            for (int b = 0; b < 4; ++b) {
                T* c = w.xb[b];
                if (c == P || c == V) continue;
                if (Nw == nullptr) Nw = c;
                else if (Vn == nullptr) Vn = c;
            }
            R coef = R(0);
            double beta_new = beta;
            if (method == DCP_LASSO_ACC_ISTA) {
                coef = (R)((double)i / (double)(i + 3));                    // lasso.py:353
            } else if (method == DCP_LASSO_FISTA) {
                beta_new = 0.5 * (1.0 + sqrt(1.0 + 4.0 * beta * beta));     // lasso.py:411
                coef = (R)((beta - 1.0) / beta_new);
            }
            const int check = (i % 10 == 0) ? 1 : 0;   // (w.flag is zero here: lasso_scalars_kernel, flag_publish_kernel)
            EpiProxStep<T, PROX> epi{w.yAt, V, P, Nw, mom ? Vn : (T*)nullptr, (long)K, w.scal,
                                     w.alphak, w.tolk, rowscale, coef, check, w.flag};
            const bool had_pending = pend_i >= 0;
            if (mask_ndim == 2) {
                GemmArgs<T> a1;   // T1 = (V An) o M
                a1.A = V; a1.lda = K; a1.B = w.An; a1.ldb = F; a1.M = N; a1.N = F; a1.K = K;
                a1.ext_ws = w.ext1;
                DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a1, EpiMulMask<T>{mask, F, w.T1, F})));
                GemmArgs<T> a2;   // back = T1 An^H, prox step in the epilogue
                a2.A = w.T1; a2.lda = F; a2.B = w.An; a2.ldb = F; a2.M = N; a2.N = K; a2.K = F;
                a2.conjB = true;
                a2.ext_ws = w.ext1;
                DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, a2, epi)));
            } else {
                GemmArgs<T> a;    // back = V H (H = I - Linv A A^H, see EpiProxStepH), prox step in the epilogue
                a.A = V; a.lda = K; a.B = w.AAt; a.ldb = K; a.M = N; a.N = K; a.K = K;
                a.ext_ws = w.ext2;
                // complex: the extended image of A A^H (ext2 holds nothing else during the loop) is built by
                // the first iteration and reused by the others (a product with fewer samples than atoms takes
                // the planar-rows form instead, which images V and ignores this flag)
                a.ext_ready = i > 0 && !cplx_planar_a<FORM_NN>(N, K, false, false, w.ext2);
                // float32, >= 512 atoms, at least one 128 x 128 tile per two CUs: the 8-wave 128 x 128 x 64 tile
                // (8192 x 512 x 512: 45 us against 50 us on the 64 x 64 tile for the bare product; dictionary step
                // 1.64 -> 1.62 ms).  complex64 measured slower on it and keeps the automatic choice.
                if (std::is_same<T, float>::value && K >= 512 && (long)ceil_div(N, 128) * ceil_div(K, 128) >= 128)
                    a.tile = TILE_MID;
                EpiProxStepH<T, PROX> epih{w.yAt, P, Nw, mom ? Vn : (T*)nullptr, (long)K, w.scal, w.alphak, w.tolk,
                                           coef, check, (check || mom) ? 1 : 0, w.flag};
                if (i == maxiter - 1 && method != DCP_LASSO_ACC_ISTA) {   // the iterate lasso.py:297 / :415 return
                    epih.xfinal = X;
                    epih.sdiv = w.s;
                    wrote_final = true;
                }
                DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a, epih)));
            }
            if (had_pending) {   // the check iteration before this one: was its test met?  (lasso.py:293-294)
                bool stop = false;
                DCP_TRY(resolve(&stop));
                if (stop) break;   // this iteration is discarded
            }
            if (check) {
                *reinterpret_cast<volatile int*>(host_flag) = -1;   // sentinel: nothing is in flight into it here
                hipLaunchKernelGGL(flag_publish_kernel<void>, dim3(1), dim3(64), 0, st, w.flag, host_flag);
                DCP_LAUNCH_OK(h, hipGetLastError());
                pend_i = i;
                pend_x = Nw;
            }
            lastP = P;
            lastNw = Nw;
            P = Nw;
            V = mom ? Vn : Nw;
            beta = beta_new;
        }
        if (!converged) {   // a check on the very last iteration
            bool stop = false;
            DCP_TRY(resolve(&stop));
        }
        // QUIRK: on exhaustion ista / fista return the latest iterate, acc_ista the one
        // before it (its `x0 = x0_new` sits at the top of the loop body, lasso.py:351-357).
        if (!converged) result = (method == DCP_LASSO_ACC_ISTA) ? lastP : lastNw;
        final_in_place = wrote_final && !converged;
    }

    // ---- x / s  (lasso.py:189) ----
    if (!final_in_place) {
        hipLaunchKernelGGL((col_scale_kernel<T>), dim3(grid_for((long)N * K)), dim3(256), 0, st,
                           (const T*)result, (const R*)w.s, (long)N, (long)K, 0, X);
        DCP_LAUNCH_OK(h, hipGetLastError());
    }
    if (!extra.no_final_sync) DCP_HIP_OK(h, hipStreamSynchronize(st));
    *it_out = it;
    return DCP_OK;
}

}  // namespace dcp
