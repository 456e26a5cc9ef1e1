// The sequential atom sweep of online dictionary learning, blocked so that everything that
// touches the F-long rows is a GEMM on the matrix cores.
//
// Reference (decomp/dictionary_learning.py:154-159), for k = 0..K-1 in order:
//     u_k = (B_k - A_k . D_new) / (A_kk + 1e-15) + D_new[k] ;  D_new[k] = u_k / sqrt(max(|u_k|^2, 1))
// where D_new holds the already updated atoms j < k and the old ones j >= k.  Atom k needs the
// full-row norm of every earlier atom: K dependent steps.
//
// Blocked form (block = b <= 64 consecutive atoms, t_j = D_new[j] after its update):
//   (1) P_k = (B_k - sum_{j not (in block, j < k)} A_kj Dcur_j) / den_k + D_old[k]
//           one GEMM [b x K].[K x F] with the block's strictly-lower A entries zeroed; Dcur =
//           D_new as it stands (earlier blocks new, this and later blocks old)
//       then u_k = P_k - sum_{j<k in block} w_kj t_j ,  w_kj = A_kj / den_k
//   (2) G = P P^H  (b x b Gram, split-K GEMM over F)
//   (3) in the block's own coefficient space (u_k = c_k . P, t_k = E_k . P):
//           c_k = e_k - sum_{j<k} w_kj E_j ;  |u_k|^2 = c_k G c_k^H ;  E_k = c_k / sqrt(max(|u_k|^2, 1))
//       b steps on b x b data by ONE small workgroup, in double precision
//   (4) D_new[block] = E . P   (GEMM [b x b].[b x F])
// The recursion is the same arithmetic re-associated (norms through the Gram matrix); rounding
// differs at the 1e-7 (fp32) / 1e-16 (fp64) level.  Cost per block: 2bKF + 4b^2F flops on MFMA and
// ~6 launches, independent of F in its sequential part.
#pragma once
#include "gemm.hpp"
#include "handle.hpp"
#include "kernels_small.hpp"

namespace dcp {

constexpr int kAtomBlkMax = 64;
// atoms per block: 64 for real dtypes, 32 for complex (the recursion's wide-type LDS image)
template <class T>
constexpr int atom_blk() { return scalar_traits<T>::is_complex ? 32 : 64; }

template <class T> struct wide_of;
template <> struct wide_of<float> { typedef double type; };
template <> struct wide_of<double> { typedef double type; };
template <> struct wide_of<c64> { typedef c128 type; };
template <> struct wide_of<c128> { typedef c128 type; };

DCP_HD double widen(float a) { return (double)a; }
DCP_HD double widen(double a) { return a; }
DCP_HD c128 widen(c64 a) { return c128{(double)a.re, (double)a.im}; }
DCP_HD c128 widen(c128 a) { return a; }
template <class T> DCP_HD T narrow(double a);
template <> DCP_HD float narrow<float>(double a) { return (float)a; }
template <> DCP_HD double narrow<double>(double a) { return a; }
template <class T> DCP_HD T narrow(c128 a);
template <> DCP_HD c64 narrow<c64>(c128 a) { return c64{(float)a.re, (float)a.im}; }
template <> DCP_HD c128 narrow<c128>(c128 a) { return a; }

DCP_HD double cdiv(double a, double b) { return a / b; }
DCP_HD c128 cdiv(c128 a, c128 b) {
    const double d = b.re * b.re + b.im * b.im;
    return c128{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}

// Block preparation: Ablk = rows [k0, k0+nb) of A with the in-block strictly-lower part zeroed;
// rden_i = 1 / (A_ii + 1e-15); Wl_ij = A_ij * rden_i for j < i (in block), else 0.
template <class T>
__global__ void __launch_bounds__(256) atom_prep_kernel(int k0, int nb, int K, const T* __restrict__ A,
                                                        T* __restrict__ Ablk, T* __restrict__ Wl,
                                                        T* __restrict__ rden) {
    typedef typename wide_of<T>::type WT;
    typedef real_t<WT> WR;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < (long)nb * K; e += (long)gridDim.x * 256L) {
        const int i = (int)(e / K), j = (int)(e % K);
        T v = A[(long)(k0 + i) * K + j];
        const int jl = j - k0;
        if (jl >= 0 && jl < i) v = zero_of<T>();
        Ablk[(long)i * K + j] = v;
    }
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nb * kAtomBlkMax; e += gridDim.x * 256) {
        const int i = e / kAtomBlkMax, j = e % kAtomBlkMax;
        const WT den = add(widen(A[(long)(k0 + i) * K + (k0 + i)]), from_real<WT>(WR(1.0e-15)));
        const WT rd = cdiv(from_real<WT>(WR(1)), den);
        T w = zero_of<T>();
        if (j < i) w = narrow<T>(mul(widen(A[(long)(k0 + i) * K + (k0 + j)]), rd));
        Wl[i * kAtomBlkMax + j] = w;
        if (j == 0) rden[i] = narrow<T>(rd);
    }
}

// P = (B_blk - acc) * rden + D_old[blk]      (epilogue of GEMM (1))
template <class T>
struct EpiAtomP {
    const T* B;      // + k0 * F
    const T* Dold;   // + k0 * F  (rows of D_new not yet updated)
    const T* rden;   // [nb]
    T* out;          // [nb, F]
    long ld;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        const long i = (long)r * ld + c;
        out[i] = add(mul(sub(B[i], v), rden[r]), Dold[i]);
    }
};

// Sum of one double per lane over the 64 lanes of a wave, with DPP row shifts / row broadcasts
// (VALU latency) instead of the LDS crossbar of __shfl_xor (six dependent ds_bpermute pairs are
// the longest chain of a recursion step otherwise).  Invalid / masked-out source lanes contribute
// +0.0 (bound_ctrl, old = 0).  The total lands in lane 63 and is read back as a wave-uniform value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int slo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int shi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return x + __hiloint2double(shi, slo);
}
__device__ __forceinline__ double wave_sum_f64(double x) {
    x = dpp_add_f64<0x111, 0xf>(x);   // row_shr:1
    x = dpp_add_f64<0x112, 0xf>(x);   // row_shr:2
    x = dpp_add_f64<0x114, 0xf>(x);   // row_shr:4
    x = dpp_add_f64<0x118, 0xf>(x);   // row_shr:8   -> lane 15 of every row holds its row sum
    x = dpp_add_f64<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
    x = dpp_add_f64<0x143, 0xc>(x);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}

// Step (3): ONE workgroup of 4 waves; lane i owns coefficient i, wave w takes every 4th term of
// the two k-long sums of a step (the step is instruction-issue bound in a single wave: ~600 wide
// instructions).  G (b x b, leading dim nb), Wl (leading dim kAtomBlkMax) -> E (leading dim nb).
// Everything the b dependent steps touch is staged in LDS first (coalesced).
// Two barriers per step: every wave forms c_k and E_k redundantly from the exchanged partial
// sums, so rows E_j only ever have to be visible to the wave that consumes them (j mod 4 == w,
// which is also the wave that stores them), and c_k is broadcast inside each wave through its own
// LDS row -- same-wave LDS traffic is ordered, no workgroup barrier needed for either.
template <class T>
__global__ void __launch_bounds__(256) atom_recur_kernel(int nb, const T* __restrict__ G,
                                                         const T* __restrict__ Wl, T* __restrict__ E) {
    typedef typename wide_of<T>::type WT;
    typedef real_t<WT> WR;
    constexpr int BMAX = atom_blk<T>();
    __shared__ WT sE[BMAX][BMAX + 1];   // rows E_j (wide type); row j is written and read by wave j % 4
    __shared__ T sG[BMAX][BMAX + 1];    // sG[i][i'] = G[i][i']
    __shared__ T sW[BMAX][BMAX + 1];    // sW[k][j] = w_kj
    __shared__ WT sc[4][BMAX];          // c_k, one private copy per wave
    __shared__ WT spart[4][BMAX];       // per-wave partial sums
    __shared__ WR snorm[4];
    const int tid = threadIdx.x, i = tid & 63, w = tid >> 6;
    for (int e = tid; e < nb * nb; e += 256) sG[e / nb][e % nb] = G[e];
    for (int e = tid; e < nb * BMAX; e += 256) sW[e / BMAX][e % BMAX] = Wl[(e / BMAX) * kAtomBlkMax + (e % BMAX)];
    __syncthreads();
    const int il = i < BMAX ? i : 0;
    const bool live = i < BMAX;
    for (int k = 0; k < nb; ++k) {
        // c_k[i] = delta_ik - sum_{j<k} w_kj E_j[i]      (wave w: j = w, w+4, ...)
        // (LDS operands fetched four at a time: the loops are LDS-latency bound otherwise)
        WT acc = zero_of<WT>();
        {
            int j = w;
            for (; j + 12 < k; j += 16) {
                const T w0 = sW[k][j], w1 = sW[k][j + 4], w2 = sW[k][j + 8], w3 = sW[k][j + 12];
                const WT e0 = sE[j][il], e1 = sE[j + 4][il], e2 = sE[j + 8][il], e3 = sE[j + 12][il];
                acc = madd(acc, widen(w0), e0);
                acc = madd(acc, widen(w1), e1);
                acc = madd(acc, widen(w2), e2);
                acc = madd(acc, widen(w3), e3);
            }
            for (; j < k; j += 4) acc = madd(acc, widen(sW[k][j]), sE[j][il]);
        }
        if (live) spart[w][i] = acc;
        __syncthreads();                                        // barrier 1: partial sums
        WT c = sub((i == k) ? from_real<WT>(WR(1)) : zero_of<WT>(),
                   add(add(spart[0][il], spart[1][il]), add(spart[2][il], spart[3][il])));
        if (i >= nb || i > k) c = zero_of<WT>();   // support of c_k is i <= k
        if (live) sc[w][i] = c;                    // this wave's own copy (same-wave LDS order)
        __builtin_amdgcn_wave_barrier();
        // |u_k|^2 = sum_{i,i'} c_i conj(c_i') G_ii'      (wave w: i' = w, w+4, ...)
        WT v = zero_of<WT>();
        if (i <= k && i < nb) {
            int ip = w;
            for (; ip + 12 <= k; ip += 16) {
                const WT a0 = sc[w][ip], a1 = sc[w][ip + 4], a2 = sc[w][ip + 8], a3 = sc[w][ip + 12];
                const T g0 = sG[il][ip], g1 = sG[il][ip + 4], g2 = sG[il][ip + 8], g3 = sG[il][ip + 12];
                v = madd(v, conj_of(a0), widen(g0));
                v = madd(v, conj_of(a1), widen(g1));
                v = madd(v, conj_of(a2), widen(g2));
                v = madd(v, conj_of(a3), widen(g3));
            }
            for (; ip <= k; ip += 4) v = madd(v, conj_of(sc[w][ip]), widen(sG[il][ip]));
        }
        const WR part = wave_sum_f64(real_part(mul(c, v)));
        if (i == 0) snorm[w] = part;
        __syncthreads();                                        // barrier 2: the four norm parts
        const WR tot = (snorm[0] + snorm[1]) + (snorm[2] + snorm[3]);
        const WT e = scale(c, rsqrt(tot > WR(1) ? tot : WR(1)));   // c / sqrt(max(|u_k|^2, 1))
        if (w == (k & 3) && live) sE[k][i] = e;    // read back only by this same wave (j % 4 == w)
        if (w == 0 && i < nb) E[(long)k * nb + i] = narrow<T>(e);
        // next step: spart is rewritten only after every wave has passed barrier 2 (all reads of
        // it sit between the barriers); snorm is rewritten only after barrier 1 of the next step
    }
}

template <class T>
struct AtomWs {
    T* Ablk = nullptr;   // [64, K]
    T* P = nullptr;      // [64, F]
    T* G = nullptr;      // [64, 64]
    T* E = nullptr;      // [64, 64]
    T* Wl = nullptr;     // [64, 64]
    T* rden = nullptr;   // [64]
    T* slabs = nullptr;  // split-K partials of G
    size_t slab_count = 0;
    real_t<T>* ext = nullptr;   // complex: real extended images (max(4KF, 4*64*F) reals)
};

template <class T>
inline void atom_plan(WsPlan& p, int64_t F, int64_t K) {
    p.add<T>((size_t)kAtomBlkMax * K);
    p.add<T>((size_t)kAtomBlkMax * F);
    p.add<T>((size_t)kAtomBlkMax * kAtomBlkMax);
    p.add<T>((size_t)kAtomBlkMax * kAtomBlkMax);
    p.add<T>((size_t)kAtomBlkMax * kAtomBlkMax);
    p.add<T>((size_t)kAtomBlkMax);
    p.add<T>((size_t)64 * kAtomBlkMax * kAtomBlkMax);
    if (scalar_traits<T>::is_complex) p.add<real_t<T> >((size_t)4 * (K > kAtomBlkMax ? K : kAtomBlkMax) * F);
}

template <class T>
inline int atom_carve(dcp_handle* h, AtomWs<T>& w, int64_t F, int64_t K) {
    w.Ablk = ws_alloc<T>(h, (size_t)kAtomBlkMax * K);
    w.P = ws_alloc<T>(h, (size_t)kAtomBlkMax * F);
    w.G = ws_alloc<T>(h, (size_t)kAtomBlkMax * kAtomBlkMax);
    w.E = ws_alloc<T>(h, (size_t)kAtomBlkMax * kAtomBlkMax);
    w.Wl = ws_alloc<T>(h, (size_t)kAtomBlkMax * kAtomBlkMax);
    w.rden = ws_alloc<T>(h, (size_t)kAtomBlkMax);
    w.slab_count = (size_t)64 * kAtomBlkMax * kAtomBlkMax;
    w.slabs = ws_alloc<T>(h, w.slab_count);
    if (scalar_traits<T>::is_complex) {
        w.ext = ws_alloc<real_t<T> >(h, (size_t)4 * (K > kAtomBlkMax ? K : kAtomBlkMax) * F);
        if (!w.ext) return fail(h, DCP_ERR_INTERNAL, "atom sweep workspace plan");
    }
    if (!w.Ablk || !w.P || !w.G || !w.E || !w.Wl || !w.rden || !w.slabs)
        return fail(h, DCP_ERR_INTERNAL, "atom sweep workspace plan");
    return DCP_OK;
}

#ifndef DCP_LAUNCH_OK
#define DCP_LAUNCH_OK(h, what)                                                        \
    do {                                                                              \
        hipError_t _e = (what);                                                       \
        if (_e != hipSuccess)                                                         \
            return dcp::fail((h), DCP_ERR_HIP, std::string("launch failed: ") +       \
                                                   hipGetErrorString(_e));            \
    } while (0)
#endif

// D_new (holding a copy of D on entry) <- the swept dictionary.  A [K,K], B [K,F] statistics.
template <class T>
inline int atom_sweep(dcp_handle* h, const T* A, const T* B, T* Dnew, int64_t F64, int64_t K64,
                      AtomWs<T>& w) {
    hipStream_t st = h->stream;
    const int K = (int)K64, F = (int)F64;
    constexpr int BLK = atom_blk<T>();
    for (int k0 = 0; k0 < K; k0 += BLK) {
        const int nb = (K - k0) < BLK ? (K - k0) : BLK;
        hipLaunchKernelGGL((atom_prep_kernel<T>), dim3(grid_for((long)nb * K, 64)), dim3(256), 0, st, k0,
                           nb, K, A, w.Ablk, w.Wl, w.rden);
        DCP_LAUNCH_OK(h, hipGetLastError());
        {   // (1) P = (B_blk - Ablk . D_cur) * rden + D_old[blk]
            GemmArgs<T> a;
            a.A = w.Ablk; a.lda = K; a.B = Dnew; a.ldb = F; a.M = nb; a.N = F; a.K = K;
            a.tile = TILE_SMALL;
            a.ext_ws = w.ext;
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a, EpiAtomP<T>{B + (long)k0 * F, Dnew + (long)k0 * F,
                                                              w.rden, w.P, (long)F})));
        }
        {   // (2) G = P P^H
            GemmArgs<T> g;
            g.A = w.P; g.lda = F; g.B = w.P; g.ldb = F; g.M = nb; g.N = nb; g.K = F;
            g.conjB = true;
            g.tile = TILE_SMALL;
            g.ext_ws = w.ext;
            plan_splits<FORM_NT>(g, 64, 64);
            if ((size_t)g.ksplits * nb * nb > w.slab_count) return fail(h, DCP_ERR_INTERNAL, "atom slab plan");
            DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, g, EpiSlab<T>{w.slabs, (long)nb, (long)nb * nb})));
            hipLaunchKernelGGL((reduce_slabs_kernel<T>), dim3(grid_for((long)nb * nb, 16)), dim3(256), 0, st,
                               w.slabs, (long)nb * nb, g.ksplits, (long)nb * nb, w.G);
            DCP_LAUNCH_OK(h, hipGetLastError());
        }
        // (3) the b-step recursion in coefficient space
        hipLaunchKernelGGL((atom_recur_kernel<T>), dim3(1), dim3(256), 0, st, nb, (const T*)w.G,
                           (const T*)w.Wl, w.E);
        DCP_LAUNCH_OK(h, hipGetLastError());
        {   // (4) D_new[blk] = E . P
            GemmArgs<T> a;
            a.A = w.E; a.lda = nb; a.B = w.P; a.ldb = F; a.M = nb; a.N = F; a.K = nb;
            a.tile = TILE_SMALL;
            a.ext_ws = w.ext;
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a, EpiStore<T>{Dnew + (long)k0 * F, (long)F})));
        }
    }
    return DCP_OK;
}

}  // namespace dcp
