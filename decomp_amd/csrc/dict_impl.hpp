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
// atom needs the full-row norm of the previous one): see atom_block_kernel below.
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

// The atom sweep (dictionary_learning.py:154-159) is sequential in k: atom k needs the
// full-row norm of every earlier atom.  It runs in super blocks of SB = 64 atoms:
//   (1) R = B_blk - A[blk, :] . D_cur            one MFMA GEMM per 64 atoms
//       (D_cur: rows < k0 already new, the rest old)
//   (2) ONE 512-thread workgroup (8 waves: 256 registers per thread) walks the 64 atoms in
//       sub-blocks of BR: thread t owns CC columns of F and
//       keeps dD_j = D_new[j] - D_old[j] of the sub-block's finished atoms in registers, so
//         u_k = (R_k - sum_{j<k in sub} A_kj dD_j) / (A_kk + 1e-15) + D_old[k]
//       needs no memory traffic and the row norm is a block reduction (no grid sync); after
//       a sub-block its dD is folded into the R rows of the super block still to come.
// 2 launches per 64 atoms instead of one launch (and a K-long strided contraction) per atom.
constexpr int kAtomSB = 64;

template <class T, int BR, int CC, int NTH>
__global__ void __launch_bounds__(NTH) atom_super_kernel(int k0, int ns, int K, long F,
                                                          const T* __restrict__ A,
                                                          T* __restrict__ Rs /* [ns, F] */,
                                                          T* __restrict__ Dnew) {
    typedef real_t<T> R_t;
    __shared__ T s_a[kAtomSB][BR + 1];   // A[k0 + r][k0 + i0 + j] for the rows r still to do
    constexpr int NW = NTH / 64;
    __shared__ R_t s_red[2][NW];         // per-wave partial norms, double buffered by atom parity
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i0 = 0; i0 < ns; i0 += BR) {
        const int nb = min(BR, ns - i0);
        __syncthreads();
        for (int e = tid; e < (ns - i0) * BR; e += NTH) {
            const int r = e / BR, j = e % BR;     // row i0 + r of the super block
            s_a[r][j] = (j < nb) ? A[(long)(k0 + i0 + r) * K + (k0 + i0 + j)] : zero_of<T>();
        }
        __syncthreads();
        T dD[BR][CC];   // statically indexed (loops fully unrolled): stays in registers
        // software pipeline: the residual / old-dictionary values of atom i+1 are in flight
        // while atom i is reduced (the barrier would otherwise expose their latency)
        T rcur[CC], dcur[CC], rnxt[CC], dnxt[CC];
#pragma clang loop unroll(full)
        for (int c = 0; c < CC; ++c) {
            const long f = tid + (long)NTH * c;
            rcur[c] = (f < F) ? Rs[(long)i0 * F + f] : zero_of<T>();
            dcur[c] = (f < F) ? Dnew[(long)(k0 + i0) * F + f] : zero_of<T>();
        }
#pragma clang loop unroll(full)
        for (int i = 0; i < BR; ++i) {
            if (i >= nb) {   // nb is uniform
#pragma clang loop unroll(full)
                for (int c = 0; c < CC; ++c) dD[i][c] = zero_of<T>();
                continue;
            }
            const int k = k0 + i0 + i;
            if (i + 1 < nb) {
#pragma clang loop unroll(full)
                for (int c = 0; c < CC; ++c) {
                    const long f = tid + (long)NTH * c;
                    rnxt[c] = (f < F) ? Rs[(long)(i0 + i + 1) * F + f] : zero_of<T>();
                    dnxt[c] = (f < F) ? Dnew[(long)(k + 1) * F + f] : zero_of<T>();
                }
            }
            T u[CC];
            R_t part = 0;
            // 1 / (A_kk + 1e-15) once per atom (a reciprocal-multiply instead of CC divisions:
            // this single-workgroup kernel is VALU-issue bound; <= 1 ulp from a true division)
            const T rden = div_scalar(from_real<T>(R_t(1)), add(s_a[i][i], from_real<T>(R_t(1.0e-15))));
#pragma clang loop unroll(full)
            for (int c = 0; c < CC; ++c) {
                T acc = rcur[c];
#pragma clang loop unroll(full)
                for (int j = 0; j < BR; ++j)
                    if (j < i) acc = sub(acc, mul(s_a[i][j], dD[j][c]));
                u[c] = add(mul(acc, rden), dcur[c]);           // dcur = D_old[k] (row k untouched)
                const long f = tid + (long)NTH * c;
                if (f >= F) u[c] = zero_of<T>();
                part += abs2(u[c]);
            }
            part = wave_sum(part);
            if (lane == 0) s_red[i & 1][wave] = part;
            __syncthreads();                  // the only barrier per atom
            R_t tot = (lane < NW) ? s_red[i & 1][lane] : R_t(0);
#pragma unroll
            for (int o = NW / 2; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
            tot = __shfl(tot, 0, 64);         // same summation order in every wave
            const R_t rnrm = R_t(1) / sqrt(tot > R_t(1) ? tot : R_t(1));   // normalize.py:2-10 (l2)
#pragma clang loop unroll(full)
            for (int c = 0; c < CC; ++c) {
                const long f = tid + (long)NTH * c;
                const T dn = scale(u[c], rnrm);
                dD[i][c] = sub(dn, dcur[c]);
                if (f < F) Dnew[(long)k * F + f] = dn;
                rcur[c] = rnxt[c];
                dcur[c] = dnxt[c];
            }
        }
        // fold this sub-block into the residual rows still to come (own columns only);
        // 4 rows at a time so that 4*CC loads are in flight per thread
        constexpr int RB = 4;
        for (int r0 = nb; r0 < ns - i0; r0 += RB) {
            T acc[RB][CC];
#pragma clang loop unroll(full)
            for (int q = 0; q < RB; ++q)
#pragma clang loop unroll(full)
                for (int c = 0; c < CC; ++c) {
                    const long f = tid + (long)NTH * c;
                    acc[q][c] = (f < F && r0 + q < ns - i0) ? Rs[(long)(i0 + r0 + q) * F + f]
                                                            : zero_of<T>();
                }
#pragma clang loop unroll(full)
            for (int q = 0; q < RB; ++q) {
                const int r = (r0 + q < ns - i0) ? (r0 + q) : r0;
#pragma clang loop unroll(full)
                for (int j = 0; j < BR; ++j) {
                    const T a = s_a[r][j];
#pragma clang loop unroll(full)
                    for (int c = 0; c < CC; ++c) acc[q][c] = sub(acc[q][c], mul(a, dD[j][c]));
                }
            }
#pragma clang loop unroll(full)
            for (int q = 0; q < RB; ++q)
#pragma clang loop unroll(full)
                for (int c = 0; c < CC; ++c) {
                    const long f = tid + (long)NTH * c;
                    if (f < F && r0 + q < ns - i0) Rs[(long)(i0 + r0 + q) * F + f] = acc[q][c];
                }
        }
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
    T* Rblk = nullptr;      // [64, F] residual rows of the atom super block being processed
    float* ext = nullptr;   // complex64: real extended image of D_new (4KF floats)
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
    // blocked atom sweep
    // threads per workgroup: 512 (256 registers each); 1024 for wide real dictionaries
    // (measured at F = 8192, K = 512: f32 3.5 ms vs 6.3 ms; complex64 is faster at 512: 17 vs 28 ms)
    const int nth = (F > 4096 && !scalar_traits<T>::is_complex) ? 1024 : 512;
    const int CCneed = (int)((F + nth - 1) / nth);
    if (CCneed > 16) return fail(h, DCP_ERR_UNSUPPORTED, "dictionary update: n_channels > 16384");
    // super-block height: with many columns per thread the sub-blocks are short (register
    // budget), so folding them into a tall super block re-streams its rows too often
    constexpr int CXW = scalar_traits<T>::is_complex ? 2 : 1;
    int sb = kAtomSB;
    if (CCneed * CXW > 8) sb = 32;
    if (CCneed * CXW > 16) sb = 16;
    for (int k0 = 0; k0 < (int)K; k0 += sb) {
        const int ns = ((int)K - k0) < sb ? ((int)K - k0) : sb;
        {   // R = B_blk - A[blk, :] . D_cur
            GemmArgs<T> a;
            a.A = A + (long)k0 * K; a.lda = K; a.B = Dnew; a.ldb = F;
            a.M = ns; a.N = (int)F; a.K = (int)K;
            a.tile = TILE_SMALL;
            a.ext_ws = w.ext;
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a, EpiSubFrom<T>{B + (long)k0 * F, (long)F, w.Rblk, (long)F})));
        }
        // sub-block height: BR x CC (x2 for complex) dD registers per thread <= 64
#define DCP_ATOM_LAUNCH(BRV, CCV, NTV)                                                              \
    hipLaunchKernelGGL((atom_super_kernel<T, BRV, CCV, NTV>), dim3(1), dim3(NTV), 0, st, k0, ns,    \
                       (int)K, (long)F, (const T*)A, w.Rblk, Dnew)
        if (nth == 512) {
            if constexpr (scalar_traits<T>::is_complex) {
                if (CCneed <= 1) DCP_ATOM_LAUNCH(32, 1, 512);
                else if (CCneed <= 2) DCP_ATOM_LAUNCH(16, 2, 512);
                else if (CCneed <= 4) DCP_ATOM_LAUNCH(8, 4, 512);
                else if (CCneed <= 8) DCP_ATOM_LAUNCH(4, 8, 512);
                else DCP_ATOM_LAUNCH(2, 16, 512);
            } else {
                if (CCneed <= 1) DCP_ATOM_LAUNCH(32, 1, 512);
                else if (CCneed <= 2) DCP_ATOM_LAUNCH(32, 2, 512);
                else if (CCneed <= 4) DCP_ATOM_LAUNCH(16, 4, 512);
                else DCP_ATOM_LAUNCH(8, 8, 512);
            }
        } else {   // 1024 threads: 128 registers each -> dD budget 32 (real types only)
            if constexpr (!scalar_traits<T>::is_complex) {
                if (CCneed <= 8) DCP_ATOM_LAUNCH(4, 8, 1024);
                else DCP_ATOM_LAUNCH(2, 16, 1024);
            }
        }
#undef DCP_ATOM_LAUNCH
        DCP_LAUNCH_OK(h, hipGetLastError());
    }
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
