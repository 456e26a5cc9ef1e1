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
// atom needs the full-row norm of the previous one): blocked so that all F-long work is GEMMs
// (atom_sweep.hpp).
#pragma once
#include "atom_sweep.hpp"
#include "atom_fused_f32.hpp"
#include "atom_fused_c64.hpp"
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
                                                              T* __restrict__ B, unsigned int* __restrict__ zero4) {
    const long W = F + K;
    const long n = K * W;
    if (blockIdx.x == 0 && threadIdx.x < 4) zero4[threadIdx.x] = 0u;   // gmax + ticket of maxabsdiff_publish_kernel
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const long r = i / W, c = i - r * W;
        const T v = stats[i];
        if (c < F) B[r * F + c] = add(scale(B[r * F + c], beta), v);
        else A[r * K + (c - F)] = add(scale(A[r * K + (c - F)], beta), v);
    }
}

// The same with the statistics still in S ordered split-K partials (one GPU: nothing has to be all-reduced, so the
// slab sum and the accumulation are one pass; same left-to-right order as reduce_slabs_kernel, hence the same bits).
template <class T>
__global__ void __launch_bounds__(256) dict_accumulate_slabs_kernel(const T* __restrict__ slabs, long stride, int S,
                                                                    long K, long F, real_t<T> beta,
                                                                    T* __restrict__ A, T* __restrict__ B,
                                                                    const T* __restrict__ Dsrc, T* __restrict__ Ddst,
                                                                    unsigned int* __restrict__ zero4) {
    const long W = F + K;
    const long n = K * W;
    if (blockIdx.x == 0 && threadIdx.x < 4) zero4[threadIdx.x] = 0u;   // gmax + ticket of maxabsdiff_publish_kernel
    // (the sweep works in place on a copy of D: the copy rides along, one launch fewer)
    if (Dsrc != nullptr)
        for (long i = blockIdx.x * 256L + threadIdx.x; i < K * F; i += (long)gridDim.x * 256L) Ddst[i] = Dsrc[i];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        T v = slabs[i];
        int s = 1;
        for (; s + 7 < S; s += 8) {
            T u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) u[q] = slabs[(long)(s + q) * stride + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) v = add(v, u[q]);
        }
        for (; s < S; ++s) v = add(v, slabs[(long)s * stride + i]);
        const long r = i / W, c = i - r * W;
        if (c < F) B[r * F + c] = add(scale(B[r * F + c], beta), v);
        else A[r * K + (c - F)] = add(scale(A[r * K + (c - F)], beta), v);
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

// max_i |a_i - b_i| -> *out in ONE launch: a returning atomic max per workgroup (|.| >= 0: the bit pattern is monotone,
// a NaN wins as in np.max), then an arrival ticket whose increment depends on the returned value; the workgroup that
// arrives last stores the finished maximum (`out` may be device-mapped pinned host memory the caller polls).
// scratch[0] (gmax) and the ticket behind it must be zero on entry: the step's accumulate kernel clears them.
// (Few, large workgroups: the two atomics of every workgroup arrive in one burst at the end and same-address atomics
// serialise at the memory side -- 512 workgroups of 256 threads spent 20 us here, 13 of them queueing.)
template <class T>
__global__ void __launch_bounds__(1024) maxabsdiff_publish_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                                  long n, real_t<T>* __restrict__ gmax,
                                                                  unsigned int* __restrict__ ticket,
                                                                  real_t<T>* __restrict__ out) {
    typedef real_t<T> R;
    __shared__ R sh[16];
    R m = 0;
    const long stride = (long)gridDim.x * 1024L;
    long i = blockIdx.x * 1024L + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        T va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { va[u] = a[i + u * stride]; vb[u] = b[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const R d = absval(sub(va[u], vb[u]));
            m = (d > m || d != d) ? d : m;
        }
    }
    for (; i < n; i += stride) {
        const R d = absval(sub(a[i], b[i]));
        m = (d > m || d != d) ? d : m;
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        R r = sh[0];
        for (int w = 1; w < 16; ++w) r = (sh[w] > r || sh[w] != sh[w]) ? sh[w] : r;
        unsigned int inc = 1u;
        R old = atomic_max_nonneg_ret(gmax, r);
        asm volatile("; the arrival is counted behind the max" : "+v"(inc) : "v"(old));
        if (atomicAdd(ticket, inc) == gridDim.x - 1u) *out = atomic_read_nonneg(gmax);
    }
}

// ---- masked dictionary learning (dictionary_learning.py:171-231) ----------------------------
// A3[k, f, j] <- beta A3 + sum_n conj(x[n,k]) x[n,j] m[n,f]      (:209-213)
// One workgroup per channel f; threads walk the (k, j) pairs.  O(F K^2 Nb): the reference's own
// formulation of this statistic; a parity path, not a tuned one.
template <class T>
__global__ void __launch_bounds__(256) dict_mask_gram_kernel(const T* __restrict__ X,
                                                             const real_t<T>* __restrict__ M, long Nb,
                                                             long F, int K, real_t<T> beta,
                                                             T* __restrict__ A3) {
    const long f = blockIdx.x;
    for (int e = threadIdx.x; e < K * K; e += 256) {
        const int k = e / K, j = e % K;
        T acc = zero_of<T>();
        for (long n = 0; n < Nb; ++n)
            acc = madd(acc, conj_of(X[n * K + k]), scale(X[n * K + j], M[n * F + f]));
        T* dst = A3 + ((long)k * F + f) * K + j;
        *dst = add(scale(*dst, beta), acc);
    }
}

// out = beta * out + in
template <class T>
__global__ void __launch_bounds__(256) scale_add_kernel(long n, real_t<T> beta, const T* __restrict__ in,
                                                        T* __restrict__ out) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L)
        out[i] = add(scale(out[i], beta), in[i]);
}

// One workgroup per atom k (they are independent: the contraction uses the OLD dictionary,
// dictionary_learning.py:219-223):  AkD_f = sum_j A3[k,f,j] D[j,f] ;  Akk = sum_f (A3[k,f,k] + eps) ;
// u = (B_k - AkD) / Akk + D_k ;  D_new[k] = u / sqrt(max(|u|^2, 1)).
template <class T>
__global__ void __launch_bounds__(256) dict_mask_atom_kernel(const T* __restrict__ A3,
                                                             const T* __restrict__ B,
                                                             const T* __restrict__ D, long F, int K,
                                                             T* __restrict__ Dnew) {
    typedef real_t<T> R;
    __shared__ R sh_re[4], sh_im[4];
    __shared__ T s_akk;
    __shared__ R s_nrm;
    const int k = blockIdx.x;
    R are = 0, aim = 0;
    for (long f = threadIdx.x; f < F; f += 256) {
        const T v = add(A3[((long)k * F + f) * K + k], from_real<T>(R(1.0e-15)));
        are += real_part(v);
        if constexpr (scalar_traits<T>::is_complex) aim += v.im;
    }
    const R tre = block_sum_256(are, sh_re);
    R tim = 0;
    if constexpr (scalar_traits<T>::is_complex) tim = block_sum_256(aim, sh_im);
    if (threadIdx.x == 0) {
        if constexpr (scalar_traits<T>::is_complex) { s_akk.re = tre; s_akk.im = tim; }
        else s_akk = tre;
    }
    __syncthreads();
    const T akk = s_akk;
    R part = 0;
    for (long f = threadIdx.x; f < F; f += 256) {
        T dot = zero_of<T>();
        const T* a = A3 + ((long)k * F + f) * K;
        for (int j = 0; j < K; ++j) dot = madd(dot, a[j], D[(long)j * F + f]);
        const T u = add(div_scalar(sub(B[(long)k * F + f], dot), akk), D[(long)k * F + f]);
        Dnew[(long)k * F + f] = u;
        part += abs2(u);
    }
    const R tot = block_sum_256(part, sh_re);
    if (threadIdx.x == 0) s_nrm = sqrt(tot > R(1) ? tot : R(1));
    __syncthreads();
    const R nrm = s_nrm;
    for (long f = threadIdx.x; f < F; f += 256) {
        T u = Dnew[(long)k * F + f];
        if constexpr (scalar_traits<T>::is_complex) { u.re = u.re / nrm; u.im = u.im / nrm; }
        else u = u / nrm;
        Dnew[(long)k * F + f] = u;
    }
}

template <class T>
struct DictWs {
    typedef real_t<T> R;
    T* slabs = nullptr;
    size_t slab_count = 0;
    R* partial = nullptr;   // max|dD| partials
    R* scal = nullptr;
    int stat_nslabs = 0;    // set by dict_local_stats
    AtomWs<T> atom;         // the blocked atom sweep's buffers (atom_sweep.hpp)
    // float32 with K or F off the 64-grid: zero-padded copies [Kp,Kp], [Kp,Fp], [Kp,Fp] for the fused sweep
    T* padA = nullptr;
    T* padB = nullptr;
    T* padD = nullptr;
};

// float32 dictionaries of more than one 64-atom block whose K or F is not a multiple of 64 run the fused
// sweep on zero-padded copies: a padded atom has A_k = 0, B_k = 0, D_k = 0, so u_k = 0 / 1e-15 + 0 = 0 and it
// stays zero without touching any other atom; padded channels are zero columns of B and D and add nothing to
// any norm.  Exactly the sweep of the unpadded problem, at the three-launches-per-block rate.
template <class T>
inline bool dict_pads(int64_t F, int64_t K) {
    return std::is_same<T, float>::value && K > 64 && !atom_fused_ok(F, K);
}
inline int64_t pad64(int64_t v) { return (v + 63) / 64 * 64; }

template <class T>
inline size_t dict_slab_elems(int64_t Nb, int64_t F, int64_t K) {
    GemmArgs<T> a;
    a.M = (int)K; a.N = (int)(F + K); a.K = (int)Nb;
    a.B2 = reinterpret_cast<const T*>(1); a.n_b1 = (int)F;   // two segments (tile count only)
    a.conjA = true;                                          // as dict_local_stats plans it
    plan_splits<FORM_TN>(a, kSplitTarget, kMaxSplits);
    GemmArgs<T> b;   // the masked variant's x^H (y o m) alone (dict_api.hpp): its own plan
    b.M = (int)K; b.N = (int)F; b.K = (int)Nb;
    b.conjA = true;
    plan_splits<FORM_TN>(b, kSplitTarget, kMaxSplits);
    const size_t both = (size_t)a.ksplits * K * (F + K), single = (size_t)b.ksplits * K * F;
    return both > single ? both : single;
}

// stats[K, F+K] = X^H [Y | X] for this rank's rows of the minibatch
// keep_slabs: leave the ordered split-K partials in w.slabs (w.stat_nslabs of them, K (F + K) apart) for
// dict_update to sum while it accumulates; `stats` is then neither read nor written (pass nullptr).
template <class T>
inline int dict_local_stats(dcp_handle* h, const T* Y, const T* X, int64_t Nb, int64_t F, int64_t K,
                            T* stats, DictWs<T>& w, bool keep_slabs = false) {
    GemmArgs<T> a;
    a.A = X; a.lda = K; a.B = Y; a.ldb = F; a.B2 = X; a.ldb2 = K; a.n_b1 = (int)F;
    a.M = (int)K; a.N = (int)(F + K); a.K = (int)Nb;
    a.conjA = true;                                          // x^H (dictionary_learning.py:147-149)
    plan_splits<FORM_TN>(a, kSplitTarget, kMaxSplits);
    const long W = F + K;
    if ((size_t)a.ksplits * K * W > w.slab_count) return fail(h, DCP_ERR_INTERNAL, "dict slab plan");
    DCP_LAUNCH_OK(h, (gemm<FORM_TN>(h->stream, a, EpiSlab<T>{w.slabs, W, (long)K * W})));
    w.stat_nslabs = a.ksplits;
    if (keep_slabs) return DCP_OK;
    if (!stats) return fail(h, DCP_ERR_INTERNAL, "dict_local_stats: stats is null");
    launch_reduce_slabs<T>(h->stream, w.slabs, (long)K * W, a.ksplits, (long)K * W, stats);
    DCP_LAUNCH_OK(h, hipGetLastError());
    return DCP_OK;
}

// A,B accumulation + atom sweep + max|D - D_new| (device scalar)
template <class T>
inline int dict_update(dcp_handle* h, const T* stats, real_t<T> beta, T* A, T* B, const T* D, T* Dnew,
                       int64_t F, int64_t K, real_t<T>* maxdiff_dev, DictWs<T>& w, int stats_nslabs = 0) {
    typedef real_t<T> R;
    hipStream_t st = h->stream;
    if (stats_nslabs > 0) {  // `stats` = ordered split-K partials (one GPU): sum and accumulate in one pass
        hipLaunchKernelGGL((dict_accumulate_slabs_kernel<T>), dim3(grid_for((long)K * (F + K))), dim3(256), 0, st,
                           stats, (long)K * (F + K), stats_nslabs, (long)K, (long)F, beta, A, B, D, Dnew,
                           reinterpret_cast<unsigned int*>(w.partial));
        DCP_LAUNCH_OK(h, hipGetLastError());
    } else {
        hipLaunchKernelGGL((dict_accumulate_kernel<T>), dim3(grid_for((long)K * (F + K))), dim3(256), 0, st,
                           stats, (long)K, (long)F, beta, A, B, reinterpret_cast<unsigned int*>(w.partial));
        DCP_LAUNCH_OK(h, hipGetLastError());
        DCP_HIP_OK(h, hipMemcpyAsync(Dnew, D, sizeof(T) * (size_t)K * F, hipMemcpyDeviceToDevice, st));
    }
    // blocked atom sweep (float32 with K, F multiples of 64: the fused three-launch path)
    bool fused = false;
    // a registered row prefetch (dcp_dict_prefetch_rows_bytes) runs on the side stream beside the sweep, whose
    // critical path -- one recursion workgroup per block -- does not touch HBM.  (Started behind the sweep's block-0
    // products instead: the same end-to-end time within noise, 1.555-1.596 against 1.558-1.565 ms per step.)
    DCP_TRY(start_registered_prefetch(h));
    if constexpr (std::is_same<T, float>::value) {
        fused = atom_fused_ok(F, K);
        if (fused) {
            DCP_TRY(atom_sweep_fused_f32(h, A, B, Dnew, F, K, w.atom));
        } else if (dict_pads<T>(F, K) && w.padA && w.padB && w.padD) {
            const int64_t Kp = pad64(K), Fp = pad64(F);
            DCP_HIP_OK(h, hipMemsetAsync(w.padA, 0, sizeof(T) * (size_t)Kp * Kp, st));
            DCP_HIP_OK(h, hipMemsetAsync(w.padB, 0, sizeof(T) * (size_t)Kp * Fp, st));
            DCP_HIP_OK(h, hipMemsetAsync(w.padD, 0, sizeof(T) * (size_t)Kp * Fp, st));
            DCP_HIP_OK(h, hipMemcpy2DAsync(w.padA, sizeof(T) * Kp, A, sizeof(T) * K, sizeof(T) * K, K,
                                           hipMemcpyDeviceToDevice, st));
            DCP_HIP_OK(h, hipMemcpy2DAsync(w.padB, sizeof(T) * Fp, B, sizeof(T) * F, sizeof(T) * F, K,
                                           hipMemcpyDeviceToDevice, st));
            DCP_HIP_OK(h, hipMemcpy2DAsync(w.padD, sizeof(T) * Fp, Dnew, sizeof(T) * F, sizeof(T) * F, K,
                                           hipMemcpyDeviceToDevice, st));
            DCP_TRY(atom_sweep_fused_f32(h, w.padA, w.padB, w.padD, Fp, Kp, w.atom));
            DCP_HIP_OK(h, hipMemcpy2DAsync(Dnew, sizeof(T) * F, w.padD, sizeof(T) * Fp, sizeof(T) * F, K,
                                           hipMemcpyDeviceToDevice, st));
            fused = true;
        }
    }
    if constexpr (std::is_same<T, c64>::value) {
        static const bool generic_only = getenv("DCP_ATOM_GENERIC") != nullptr;    // analysis knob
        if (!generic_only && atom_fused_c64_ok(F, K) && w.atom.fused_reals != nullptr) {
            DCP_TRY(atom_sweep_fused_c64(h, A, B, Dnew, F, K, w.atom));
            fused = true;
        }
    }
    if (!fused) DCP_TRY(atom_sweep<T>(h, A, B, Dnew, F, K, w.atom));
    // max|D - D_new| in one launch (w.partial: 16 bytes = the running maximum + the arrival ticket, cleared by the
    // accumulate kernel above)
    long mb = ((long)K * F + 16 * 1024 - 1) / (16 * 1024);   // ~16 elements per thread
    mb = mb < 1 ? 1 : (mb > 128 ? 128 : mb);
    hipLaunchKernelGGL((maxabsdiff_publish_kernel<T>), dim3((unsigned)mb), dim3(1024), 0, st, D, (const T*)Dnew, (long)K * F,
                       w.partial, reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(w.partial) + 8), maxdiff_dev);
    DCP_LAUNCH_OK(h, hipGetLastError());
    return DCP_OK;
}

}  // namespace dcp
