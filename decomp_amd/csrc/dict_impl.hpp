// Online dictionary learning, block coordinate descent (Mairal et al.) on the device.
//
// Reference path restated as kernels:
//   decomp/dictionary_learning.py:135-164  one minibatch step of solve_cd:
//     lasso (solve_fastpath)  ->  A = beta A + x^H x ; B = beta B + x^H y  ->
//     sequential atom sweep  u_k = (B_k - A_k . D_new) / (A_kk + 1e-15) + D_new[k],
//     D_new[k] = u_k / sqrt(max(|u_k|^2, 1))  ->  max |D - D_new|
//
// The statistics product x^H [y | x] is one split-K GEMM (the same [K, F+K] layout that the
// data-parallel driver all-reduces).  The atom sweep is inherently sequential in k (every
// atom needs the full-row norm of the previous one): it runs as K+1 small launches, each
// finalising atom k-1 (norm from per-workgroup partials) and forming the un-normalised atom
// k for a 64-column stripe per workgroup, 16 waves splitting the K-long contraction.
#pragma once
#include "lasso_impl.hpp"

namespace dcp {

// (a / b) for the update's (B_k - A_k.D) / (A_kk + eps); complex division when T is complex.
DCP_HD float div_scalar(float a, float b) { return a / b; }
DCP_HD double div_scalar(double a, double b) { return a / b; }
template <class R>
DCP_HD cx<R> div_scalar(cx<R> a, cx<R> b) {
    const R den = b.re * b.re + b.im * b.im;
    return cx<R>{(a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den};
}

// stats[K, F+K] = sum of slabs; B <- beta B + stats[:, :F] ; A <- beta A + stats[:, F:]
template <class T>
__global__ void __launch_bounds__(256) dict_accumulate_kernel(const T* __restrict__ stats, long K, long F,
                                                              real_t<T> beta, T* __restrict__ A,
                                                              T* __restrict__ B) {
    const long W = F + K;
    const long n = K * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const long r = i / W, c = i - r * W;
        const T v = stats[i];
        if (c < F) B[r * F + c] = add(scale(B[r * F + c], beta), v);
        else A[r * K + (c - F)] = add(scale(A[r * K + (c - F)], beta), v);
    }
}

// One launch per atom (k = 0..K; the last one only finalises atom K-1).
// grid = ceil(F / 64) workgroups of 1024 threads (16 waves); lane = column.
template <class T>
__global__ void __launch_bounds__(1024) atom_step_kernel(int k, int K, long F, const T* __restrict__ A,
                                                         const T* __restrict__ B, T* __restrict__ Dnew,
                                                         real_t<T>* __restrict__ partial /* [2][grid] */) {
    typedef real_t<T> R;
    __shared__ T s_part[16][64];
    __shared__ T s_fresh[64];
    __shared__ R s_red[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long f = (long)blockIdx.x * 64 + lane;
    const bool fok = f < F;
    const int G = gridDim.x;

    // ---- finalise atom k-1: D_new[k-1] = u / sqrt(max(|u|^2, 1))   (normalize.py:2-10) ----
    if (k > 0) {
        if (wave == 0) {
            R tot = 0;
            const R* p = partial + (long)((k - 1) & 1) * G;
            for (int g = 0; g < G; ++g) tot += p[g];      // same order in every workgroup
            const R nrm = sqrt(tot > R(1) ? tot : R(1));
            T u = zero_of<T>();
            if (fok) {
                u = Dnew[(long)(k - 1) * F + f];
                if constexpr (scalar_traits<T>::is_complex) { u.re = u.re / nrm; u.im = u.im / nrm; }
                else u = u / nrm;
                Dnew[(long)(k - 1) * F + f] = u;
            }
            s_fresh[lane] = u;
        }
    }
    __syncthreads();
    if (k >= K) return;

    // ---- u_k = (B_k - A_k . D_new) / (A_kk + 1e-15) + D_new[k]   (dictionary_learning.py:156) ----
    T acc = zero_of<T>();
    const T* arow = A + (long)k * K;
    for (int j = wave; j < K; j += 16) {
        T d;
        if (j == k - 1) d = s_fresh[lane];            // just normalised above (not yet re-read)
        else d = fok ? Dnew[(long)j * F + f] : zero_of<T>();
        acc = madd(acc, arow[j], d);
    }
    s_part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        T dot = s_part[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) dot = add(dot, s_part[w][lane]);
        T u = zero_of<T>();
        if (fok) {
            const T dk = Dnew[(long)k * F + f];   // still D[k]: row k is untouched so far
            const T den = add(arow[k], from_real<T>(R(1.0e-15)));
            u = add(div_scalar(sub(B[(long)k * F + f], dot), den), dk);
            Dnew[(long)k * F + f] = u;
        }
        R v = fok ? abs2(u) : R(0);
        v = wave_sum(v);
        if (lane == 0) partial[(long)(k & 1) * G + blockIdx.x] = v;
    }
}

// out[i, :] = in[index[i], :]   (MinibatchData.shuffle / .array: utils/data.py:147-156)
template <class T>
__global__ void __launch_bounds__(256) gather_rows_kernel(const T* __restrict__ in, const long long* __restrict__ index,
                                                          long rows, long cols, T* __restrict__ out) {
    const long n = rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const long r = i / cols, c = i - r * cols;
        out[i] = in[(long)index[r] * cols + c];
    }
}

// max_i |a_i - b_i| -> partial[block]  (then final_max_kernel)
template <class T>
__global__ void __launch_bounds__(256) maxabsdiff_partial_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                                 long n, real_t<T>* __restrict__ partial) {
    typedef real_t<T> R;
    __shared__ R sh[4];
    R m = 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const R d = absval(sub(a[i], b[i]));
        m = (d > m || d != d) ? d : m;
    }
    R r = block_max_256(m, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

template <class T>
struct DictWs {
    typedef real_t<T> R;
    T* slabs = nullptr;
    size_t slab_count = 0;
    R* partial = nullptr;   // [2][G] atom norms / max partials
    R* scal = nullptr;
};

template <class T>
inline size_t dict_slab_elems(int64_t Nb, int64_t F, int64_t K) {
    GemmArgs<T> a;
    a.M = (int)K; a.N = (int)(F + K); a.K = (int)Nb;
    a.B2 = reinterpret_cast<const T*>(1); a.n_b1 = (int)F;   // two segments (tile count only)
    plan_splits<FORM_TN>(a, kSplitTarget, kMaxSplits);
    return (size_t)a.ksplits * K * (F + K);
}

// stats[K, F+K] = X^H [Y | X] for this rank's rows of the minibatch
template <class T>
inline int dict_local_stats(dcp_handle* h, const T* Y, const T* X, int64_t Nb, int64_t F, int64_t K,
                            T* stats, DictWs<T>& w) {
    GemmArgs<T> a;
    a.A = X; a.lda = K; a.B = Y; a.ldb = F; a.B2 = X; a.ldb2 = K; a.n_b1 = (int)F;
    a.M = (int)K; a.N = (int)(F + K); a.K = (int)Nb;
    a.conjA = true;                                          // x^H (dictionary_learning.py:147-149)
    plan_splits<FORM_TN>(a, kSplitTarget, kMaxSplits);
    const long W = F + K;
    if ((size_t)a.ksplits * K * W > w.slab_count) return fail(h, DCP_ERR_INTERNAL, "dict slab plan");
    DCP_LAUNCH_OK(h, (gemm<FORM_TN>(h->stream, a, EpiSlab<T>{w.slabs, W, (long)K * W})));
    hipLaunchKernelGGL((reduce_slabs_kernel<T>), dim3(grid_for((long)K * W)), dim3(256), 0, h->stream,
                       w.slabs, (long)K * W, a.ksplits, (long)K * W, stats);
    DCP_LAUNCH_OK(h, hipGetLastError());
    return DCP_OK;
}

// A,B accumulation + atom sweep + max|D - D_new| (device scalar)
template <class T>
inline int dict_update(dcp_handle* h, const T* stats, real_t<T> beta, T* A, T* B, const T* D, T* Dnew,
                       int64_t F, int64_t K, real_t<T>* maxdiff_dev, DictWs<T>& w) {
    typedef real_t<T> R;
    hipStream_t st = h->stream;
    hipLaunchKernelGGL((dict_accumulate_kernel<T>), dim3(grid_for((long)K * (F + K))), dim3(256), 0, st,
                       stats, (long)K, (long)F, beta, A, B);
    DCP_LAUNCH_OK(h, hipGetLastError());
    DCP_HIP_OK(h, hipMemcpyAsync(Dnew, D, sizeof(T) * (size_t)K * F, hipMemcpyDeviceToDevice, st));
    const int G = (int)((F + 63) / 64);
    for (int k = 0; k <= (int)K; ++k) {
        hipLaunchKernelGGL((atom_step_kernel<T>), dim3(G), dim3(1024), 0, st, k, (int)K, (long)F,
                           (const T*)A, (const T*)B, Dnew, w.partial);
    }
    DCP_LAUNCH_OK(h, hipGetLastError());
    const int mb = grid_for((long)K * F, 256);
    hipLaunchKernelGGL((maxabsdiff_partial_kernel<T>), dim3(mb), dim3(256), 0, st, D, (const T*)Dnew,
                       (long)K * F, w.partial);
    DCP_LAUNCH_OK(h, hipGetLastError());
    hipLaunchKernelGGL((final_max_kernel<R>), dim3(1), dim3(256), 0, st, (const R*)w.partial, (long)mb,
                       maxdiff_dev);
    DCP_LAUNCH_OK(h, hipGetLastError());
    return DCP_OK;
}

}  // namespace dcp
