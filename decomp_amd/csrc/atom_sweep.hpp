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
//       b steps on b x b data by ONE small workgroup, in double precision (atom_recur_kernel)
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

// Preparation of ALL blocks in one launch (blk atoms per block): Ablk = A with every block's
// in-block strictly-lower part zeroed; rden_r = 1 / (A_rr + 1e-15); Wl[r][j] = A[r][k0 + j] * rden_r for
// j < r - k0 (k0 = first atom of r's block), else 0.
// Look-ahead images: Alook = Ablk with the PREVIOUS block's columns zeroed as well (the part of P that
// does not wait for that block), Aprev[r][j] = -rden_r * A[r][k0 - blk + j] (its contribution, applied
// once the previous block's atoms exist).
template <class T>
__global__ void __launch_bounds__(256) atom_prep_kernel(int blk, int K, const T* __restrict__ A,
                                                        T* __restrict__ Ablk, T* __restrict__ Wl,
                                                        T* __restrict__ rden, T* __restrict__ Alook,
                                                        T* __restrict__ Aprev, real_t<T>* __restrict__ rows_blk,
                                                        real_t<T>* __restrict__ rows_look,
                                                        real_t<T>* __restrict__ rows_prev) {
    typedef typename wide_of<T>::type WT;
    typedef real_t<WT> WR;
    // rows_*: planar-rows images of the same matrices (row 2r = Re, row 2r+1 = Im), complex64 only: the left
    // operands of the sweep's A.B products as the MFMA core takes them (gemm.hpp, cplx_rows_kernel)
    for (long e = blockIdx.x * 256L + threadIdx.x; e < (long)K * K; e += (long)gridDim.x * 256L) {
        const int r = (int)(e / K), j = (int)(e % K);
        const int k0 = (r / blk) * blk;
        T v = A[e];
        const int jl = j - k0;
        if (jl >= 0 && jl < r - k0) v = zero_of<T>();
        Ablk[e] = v;
        if constexpr (scalar_traits<T>::is_complex)
            if (rows_blk) { rows_blk[(2L * r) * K + j] = v.re; rows_blk[(2L * r + 1) * K + j] = v.im; }
        if (jl < 0 && jl >= -blk) v = zero_of<T>();
        Alook[e] = v;
        if constexpr (scalar_traits<T>::is_complex)
            if (rows_look) { rows_look[(2L * r) * K + j] = v.re; rows_look[(2L * r + 1) * K + j] = v.im; }
    }
    for (long e = blockIdx.x * 256L + threadIdx.x; e < (long)K * kAtomBlkMax; e += (long)gridDim.x * 256L) {
        const int r = (int)(e / kAtomBlkMax), j = (int)(e % kAtomBlkMax);
        const int k0 = (r / blk) * blk;
        T v = zero_of<T>();
        if (k0 >= blk && j < blk) {
            const WT den = add(widen(A[(long)r * K + r]), from_real<WT>(WR(1.0e-15)));
            const WT rd = cdiv(from_real<WT>(WR(1)), den);
            v = narrow<T>(mul(widen(A[(long)r * K + (k0 - blk + j)]), scale(rd, WR(-1))));
        }
        Aprev[e] = v;
        if constexpr (scalar_traits<T>::is_complex)
            if (rows_prev) {
                rows_prev[(2L * r) * kAtomBlkMax + j] = v.re;
                rows_prev[(2L * r + 1) * kAtomBlkMax + j] = v.im;
            }
    }
    for (long e = blockIdx.x * 256L + threadIdx.x; e < (long)K * kAtomBlkMax; e += (long)gridDim.x * 256L) {
        const int r = (int)(e / kAtomBlkMax), j = (int)(e % kAtomBlkMax);
        const int k0 = (r / blk) * blk, i = r - k0;
        const WT den = add(widen(A[(long)r * K + r]), from_real<WT>(WR(1.0e-15)));
        const WT rd = cdiv(from_real<WT>(WR(1)), den);
        T w = zero_of<T>();
        if (j < i) w = narrow<T>(mul(widen(A[(long)r * K + (k0 + j)]), rd));
        Wl[e] = w;
        if (j == 0) rden[r] = narrow<T>(rd);
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

// out += acc   (the previous block's contribution to a look-ahead P)
template <class T>
struct EpiAddTo {
    T* out;
    long ld;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        const long i = (long)r * ld + c;
        out[i] = add(out[i], v);
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

// Step (3): the b dependent steps in coefficient space.  Lane i owns coefficient i.  Besides the rows
// E_j the kernel keeps
//     M_j = G E_j^H                                    (lane i holds (G E_j^H)[i])
// so that BOTH sums of a step are lane-local over the already finished rows j < k:
//     c_k[i] = delta_ik - sum_j w_kj E_j[i]            y_k[i] = (G c_k^H)[i] = G[i][k] - sum_j conj(w_kj) M_j[i]
// and |u_k|^2 = c_k G c_k^H = Re sum_i c_k[i] y_k[i] is ONE wave reduction; then
//     E_k = c_k / n ,  M_k = y_k / n ,  n = sqrt(max(|u_k|^2, 1))     (M_k comes for free).
// Right-looking in sub-blocks of 8 atoms: rows k of the two LDS images start as (e_k, G[:, k]) and
// carry the partial sums; wave 0 runs the 8 dependent steps of a sub-block entirely in registers, then
// ALL four waves subtract that sub-block's contribution from every later row (independent rows: 8 terms
// each).  The sequential chain of a step is two FMAs, one DPP reduction and one rsqrt instead of k
// LDS-fed terms; two barriers per 8 atoms instead of two per atom.  G (b x b, leading dim nb), Wl (leading dim kAtomBlkMax) -> E (leading dim nb).
constexpr int kAtomSub = 8;

template <class T, int NTH = 256>
__device__ __forceinline__ void atom_recur_body(unsigned char* atom_lds_raw, int nb, const T* __restrict__ G,
                                                const T* __restrict__ Wl, T* __restrict__ E,
                                                real_t<T>* __restrict__ E_rows = nullptr,
                                                real_t<T>* __restrict__ E_ext = nullptr) {
    typedef typename wide_of<T>::type WT;
    typedef real_t<WT> WR;
    constexpr int BMAX = atom_blk<T>();
    // 2 x BMAX^2 wide elements + BMAX^2 of T (80 KiB for float): dynamic LDS, see atom_recur_lds_bytes
    typedef WT (*Mat)[BMAX];
    typedef T (*MatT)[BMAX];
    Mat sE = reinterpret_cast<Mat>(atom_lds_raw);                              // row k: partial c_k, then E_k
    Mat sM = reinterpret_cast<Mat>(atom_lds_raw + sizeof(WT) * BMAX * BMAX);   // row k: partial y_k, then M_k
    MatT sW = reinterpret_cast<MatT>(atom_lds_raw + 2 * sizeof(WT) * BMAX * BMAX);   // sW[k][j] = w_kj
    const int tid = threadIdx.x, i = tid & 63, w = tid >> 6;
    const bool lane_ok = i < BMAX;
    const bool live = i < nb;
    // Every global load of the prologue is issued before the first LDS store (one element per loop trip is a chain of
    // 16 memory round trips: 6.9 of the kernel's 25 us, in-kernel stamps).  sM[k][i] = G[i][k] = conj(G[k][i]): G = P P^H
    // is Hermitian bit for bit (same products, same summation order on both sides of the diagonal), so the image is
    // filled row by row -- the transposed fill wrote 64 lanes into one LDS bank.
    {
        constexpr int U = (BMAX * BMAX + NTH - 1) / NTH;
        T gv[U], wl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = tid + NTH * u;
            const int k = e / BMAX, ii = e % BMAX;
            const bool in = (k < nb && ii < nb);
            gv[u] = G[in ? k * nb + ii : 0];
            wl[u] = Wl[(k < nb ? k : 0) * kAtomBlkMax + ii];
            if (!in) gv[u] = zero_of<T>();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = tid + NTH * u;
            const int k = e / BMAX, ii = e % BMAX;
            if (k < nb) {
                sW[k][ii] = wl[u];
                sM[k][ii] = conj_of(widen(gv[u]));
                sE[k][ii] = (ii == k) ? from_real<WT>(WR(1)) : zero_of<WT>();
            }
        }
    }
    __syncthreads();
    for (int s0 = 0; s0 < nb; s0 += kAtomSub) {
        const int s1 = (s0 + kAtomSub < nb) ? s0 + kAtomSub : nb;
        if (w == 0) {
            // ---- the dependent steps of this sub-block: one wave, registers only (ALL 64 lanes run
            //      it: the DPP reduction reads lane 63).  The sub-block's pending rows and its w_kj are
            //      fetched up front; a finished row is applied to the later rows of the sub-block right
            //      away (right-looking), so the dependent chain of a step is
            //      2 FMAs -> product -> DPP reduction -> rsqrt -> 2 scalings: no LDS round trip. ----
            const int il = lane_ok ? i : 0;
            const int cnt = s1 - s0;
            WT cp[kAtomSub], yp[kAtomSub], wv[kAtomSub][kAtomSub];
#pragma unroll
            for (int t = 0; t < kAtomSub; ++t) {
                const int k = (t < cnt) ? s0 + t : s0;
                cp[t] = sE[k][il];
                yp[t] = sM[k][il];
#pragma unroll
                for (int j = 0; j < t; ++j) wv[t][j] = (t < cnt) ? widen(sW[k][s0 + j]) : zero_of<WT>();
            }
#pragma unroll
            for (int t = 0; t < kAtomSub; ++t) {
                if (t < cnt) {                                   // wave-uniform
                    const int k = s0 + t;
                    WT c = cp[t], y = yp[t];
                    if (!live || i > k) c = zero_of<WT>();        // support of c_k is i <= k (exact zeros)
                    if (!live) y = zero_of<WT>();
                    const WR tot = wave_sum_f64(real_part(mul(c, y)));
                    const WR rn = rsqrt(tot > WR(1) ? tot : WR(1));      // 1 / sqrt(max(|u_k|^2, 1))
                    const WT e = scale(c, rn);
                    const WT m = scale(y, rn);
#pragma unroll
                    for (int t2 = t + 1; t2 < kAtomSub; ++t2) {   // later rows of the sub-block
                        cp[t2] = msub(cp[t2], wv[t2][t], e);
                        yp[t2] = msub(yp[t2], conj_of(wv[t2][t]), m);
                    }
                    if (lane_ok) {
                        sE[k][i] = e;
                        sM[k][i] = m;
                    }
                    if (live) {
                        const T en = narrow<T>(e);
                        E[(long)k * nb + i] = en;
                        if constexpr (scalar_traits<T>::is_complex)
                            if (E_rows) {   // planar rows of E for the E.P product
                                E_rows[(2L * k) * nb + i] = en.re;
                                E_rows[(2L * k + 1) * nb + i] = en.im;
                            }
                        if constexpr (scalar_traits<T>::is_complex)
                            if (E_ext) {    // real left-multiplication image [2 nb, 2 nb] (atom_fused_c64.hpp)
                                real_t<T>* o = E_ext + (2L * k) * (2 * nb) + 2 * i;
                                o[0] = en.re;
                                o[1] = -en.im;
                                o[2 * nb] = en.im;
                                o[2 * nb + 1] = en.re;
                            }
                    }
                }
            }
        }
        __syncthreads();
        // ---- every later row loses this sub-block's contribution (rows are independent) ----
        if (lane_ok && s1 + w < nb) {
            // the sub-block's finished rows are the same for every later row: fetched once per wave (the compiler
            // cannot hoist them itself across the stores to sE / sM below)
            WT ej[kAtomSub], mj[kAtomSub];
#pragma unroll
            for (int t = 0; t < kAtomSub; ++t) {             // s1 - s0 == kAtomSub here (k >= s1 exists only then)
                ej[t] = sE[s0 + t][i];
                mj[t] = sM[s0 + t][i];
            }
            for (int k = s1 + w; k < nb; k += NTH / 64) {     // (NTH / 64 waves share the later rows)
                WT c = sE[k][i], y = sM[k][i];
                WT c1 = zero_of<WT>(), y1 = zero_of<WT>();
#pragma unroll
                for (int t = 0; t < kAtomSub; t += 2) {
                    const WT w0 = widen(sW[k][s0 + t]), w1 = widen(sW[k][s0 + t + 1]);
                    c = msub(c, w0, ej[t]);            c1 = msub(c1, w1, ej[t + 1]);
                    y = msub(y, conj_of(w0), mj[t]);   y1 = msub(y1, conj_of(w1), mj[t + 1]);
                }
                sE[k][i] = add(c, c1);
                sM[k][i] = add(y, y1);
            }
        }
        __syncthreads();
    }
}

template <class T>
__global__ void __launch_bounds__(256) atom_recur_kernel(int nb, const T* __restrict__ G,
                                                         const T* __restrict__ Wl, T* __restrict__ E,
                                                         real_t<T>* __restrict__ E_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char atom_lds_dyn[];
    atom_recur_body<T>(atom_lds_dyn, nb, G, Wl, E, E_rows);
}

template <class T>
constexpr size_t atom_recur_lds_bytes() {
    return (2 * sizeof(typename wide_of<T>::type) + sizeof(T)) * atom_blk<T>() * atom_blk<T>();
}

template <class T>
struct AtomWs {
    T* Ablk = nullptr;   // [K, K]   every block's rows (prepared once per sweep)
    T* Alook = nullptr;  // [K, K]   the same with the previous block's columns zeroed
    T* Aprev = nullptr;  // [K, 64]  -rden * A[block, previous block]
    T* P = nullptr;      // 2 x [64, F]  (ping-pong: block b + 1 is prepared while block b is swept)
    T* G = nullptr;      // [64, 64]
    T* E = nullptr;      // [64, 64]
    T* Wl = nullptr;     // [K, 64]
    T* rden = nullptr;   // [K]
    T* slabs = nullptr;  // split-K partials of G
    size_t slab_count = 0;
    real_t<T>* ext = nullptr;   // complex: real extended images (max(4KF, 4*64*F) reals)
    // complex: planar-rows images of Ablk, Alook [2K, K], Aprev [2K, 64] and E [128, 64]
    real_t<T>* rows_blk = nullptr;
    real_t<T>* rows_look = nullptr;
    real_t<T>* rows_prev = nullptr;
    real_t<T>* rows_E = nullptr;
    // complex64, K % 32 == 0, F % 64 == 0: the real images of the fused path (atom_fused_c64.hpp)
    real_t<T>* fused_reals = nullptr;
};

inline bool atom_fused_c64_shape(int64_t F, int64_t K) { return K >= 64 && (K % 32) == 0 && F >= 64 && (F % 64) == 0; }
inline size_t atom_fused_c64_real_count(int64_t F, int64_t K) {
    return (size_t)2 * K * F + (size_t)4 * K * K + (size_t)64 * 2 * K + (size_t)2 * K * 64 + (size_t)64 * 64 + 256;
}

// Gram slabs: up to 64 split-K slabs of the generic path, or one per 64-column tile of the fused float path
inline size_t atom_slab_elems(int64_t F) {
    const size_t n = (size_t)((F + 63) / 64);
    return (n > 64 ? n : 64) * kAtomBlkMax * kAtomBlkMax;
}

template <class T>
inline void atom_plan(WsPlan& p, int64_t F, int64_t K) {
    p.add<T>((size_t)K * K);
    p.add<T>((size_t)K * K);
    p.add<T>((size_t)K * kAtomBlkMax);
    p.add<T>((size_t)2 * kAtomBlkMax * F);
    p.add<T>((size_t)kAtomBlkMax * kAtomBlkMax);
    p.add<T>((size_t)kAtomBlkMax * kAtomBlkMax);
    p.add<T>((size_t)K * kAtomBlkMax);
    p.add<T>((size_t)K);
    p.add<T>(atom_slab_elems(F));
    if (scalar_traits<T>::is_complex) p.add<real_t<T> >((size_t)4 * (K > kAtomBlkMax ? K : kAtomBlkMax) * F);
    if (scalar_traits<T>::is_complex) {
        p.add<real_t<T> >((size_t)2 * K * K);
        p.add<real_t<T> >((size_t)2 * K * K);
        p.add<real_t<T> >((size_t)2 * K * kAtomBlkMax);
        p.add<real_t<T> >((size_t)2 * kAtomBlkMax * kAtomBlkMax);
    }
    if (std::is_same<T, c64>::value && atom_fused_c64_shape(F, K)) p.add<real_t<T> >(atom_fused_c64_real_count(F, K));
}

template <class T>
inline int atom_carve(dcp_handle* h, AtomWs<T>& w, int64_t F, int64_t K) {
    w.Ablk = ws_alloc<T>(h, (size_t)K * K);
    w.Alook = ws_alloc<T>(h, (size_t)K * K);
    w.Aprev = ws_alloc<T>(h, (size_t)K * kAtomBlkMax);
    w.P = ws_alloc<T>(h, (size_t)2 * kAtomBlkMax * F);
    w.G = ws_alloc<T>(h, (size_t)kAtomBlkMax * kAtomBlkMax);
    w.E = ws_alloc<T>(h, (size_t)kAtomBlkMax * kAtomBlkMax);
    w.Wl = ws_alloc<T>(h, (size_t)K * kAtomBlkMax);
    w.rden = ws_alloc<T>(h, (size_t)K);
    w.slab_count = atom_slab_elems(F);
    w.slabs = ws_alloc<T>(h, w.slab_count);
    if (scalar_traits<T>::is_complex) {
        w.ext = ws_alloc<real_t<T> >(h, (size_t)4 * (K > kAtomBlkMax ? K : kAtomBlkMax) * F);
        if (!w.ext) return fail(h, DCP_ERR_INTERNAL, "atom sweep workspace plan");
    }
    if (scalar_traits<T>::is_complex) {
        w.rows_blk = ws_alloc<real_t<T> >(h, (size_t)2 * K * K);
        w.rows_look = ws_alloc<real_t<T> >(h, (size_t)2 * K * K);
        w.rows_prev = ws_alloc<real_t<T> >(h, (size_t)2 * K * kAtomBlkMax);
        w.rows_E = ws_alloc<real_t<T> >(h, (size_t)2 * kAtomBlkMax * kAtomBlkMax);
        if (!w.rows_blk || !w.rows_look || !w.rows_prev || !w.rows_E)
            return fail(h, DCP_ERR_INTERNAL, "atom sweep workspace plan");
    }
    if (std::is_same<T, c64>::value && atom_fused_c64_shape(F, K)) {
        w.fused_reals = ws_alloc<real_t<T> >(h, atom_fused_c64_real_count(F, K));
        if (!w.fused_reals) return fail(h, DCP_ERR_INTERNAL, "atom sweep workspace plan");
    }
    if (!w.Ablk || !w.Alook || !w.Aprev || !w.P || !w.G || !w.E || !w.Wl || !w.rden || !w.slabs)
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
    hipLaunchKernelGGL((atom_prep_kernel<T>), dim3(grid_for((long)K * K, 256)), dim3(256), 0, st, BLK, K, A,
                       w.Ablk, w.Wl, w.rden, w.Alook, w.Aprev, w.rows_blk, w.rows_look, w.rows_prev);
    DCP_LAUNCH_OK(h, hipGetLastError());
    {
        static DynLdsRaised raised;   // per dtype
        std::atomic<bool>& r = raised.on_current_device();
        if (!r) {
            DCP_LAUNCH_OK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&atom_recur_kernel<T>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)atom_recur_lds_bytes<T>()));
            r = true;
        }
    }
    // The 64-row products of a block run on few CUs and are latency bound: 64-deep K blocks
    // (TILE_SMALL_DEEP) put 4x more loads in flight per barrier than the 16-deep small tile.
    //
    // Look-ahead: the recursion of block b is ONE workgroup, so the K-deep part of block b + 1's P
    // (everything except block b's own atoms, which do not exist yet) runs on the side stream beside
    // it; once D_new[block b] is there, its contribution is a 64-deep product added in place.
    const int nblk = (K + BLK - 1) / BLK;
    const bool lookahead = nblk > 1;
    auto p_product = [&](hipStream_t s_, const T* Arows, const real_t<T>* Aplanar, int k0, int nb, T* Pout) -> int {
        GemmArgs<T> a;   // P = (B_blk - Arows . D_cur) * rden + D_old[blk]
        a.A = Arows + (long)k0 * K; a.lda = K; a.B = Dnew; a.ldb = F; a.M = nb; a.N = F; a.K = K;
        if (Aplanar) { a.A_rows = Aplanar + 2L * k0 * K; a.lda_rows = K; }
        a.tile = TILE_SMALL_DEEP;
        a.ext_ws = w.ext;
        DCP_LAUNCH_OK(h, (gemm<FORM_NN>(s_, a, EpiAtomP<T>{B + (long)k0 * F, Dnew + (long)k0 * F, w.rden + k0,
                                                           Pout, (long)F})));
        return DCP_OK;
    };
    {   // block 0: nothing to wait for
        const int nb0 = K < BLK ? K : BLK;
        DCP_TRY(p_product(st, w.Ablk, w.rows_blk, 0, nb0, w.P));
    }
    for (int b = 0; b < nblk; ++b) {
        const int k0 = b * BLK;
        const int nb = (K - k0) < BLK ? (K - k0) : BLK;
        T* P = w.P + (size_t)(b & 1) * kAtomBlkMax * F;
        T* Pnext = w.P + (size_t)((b + 1) & 1) * kAtomBlkMax * F;
        const bool has_next = (b + 1) < nblk;
        const int k1 = k0 + BLK;
        const int nb1 = has_next ? ((K - k1) < BLK ? (K - k1) : BLK) : 0;
        {   // (2) G = P P^H : one 64-deep K block per split
            GemmArgs<T> g;
            g.A = P; g.lda = F; g.B = P; g.ldb = F; g.M = nb; g.N = nb; g.K = F;
            g.conjB = true;
            g.tile = TILE_SMALL_DEEP;
            g.ext_ws = w.ext;
            plan_splits<FORM_NT>(g, 64, 64, 4);
            if ((size_t)g.ksplits * nb * nb > w.slab_count) return fail(h, DCP_ERR_INTERNAL, "atom slab plan");
            DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, g, EpiSlab<T>{w.slabs, (long)nb, (long)nb * nb})));
            hipLaunchKernelGGL((reduce_slabs_kernel<T>), dim3(grid_for((long)nb * nb, 16)), dim3(256), 0, st,
                               w.slabs, (long)nb * nb, g.ksplits, (long)nb * nb, w.G);
            DCP_LAUNCH_OK(h, hipGetLastError());
        }
        if (has_next && lookahead) {
            // side stream: P_next without block b's atoms.  It starts after everything enqueued so far
            // (blocks < b are final in D_new; the complex path's ext scratch is free again) and reads
            // block b's OLD rows only through zero weights.
            DCP_TRY(side_after_main(h));
            DCP_TRY(p_product(h->side, w.Alook, w.rows_look, k1, nb1, Pnext));
        }
        // (3) the b-step recursion in coefficient space
        hipLaunchKernelGGL((atom_recur_kernel<T>), dim3(1), dim3(256), atom_recur_lds_bytes<T>(), st, nb,
                           (const T*)w.G, (const T*)(w.Wl + (long)k0 * kAtomBlkMax), w.E, w.rows_E);
        DCP_LAUNCH_OK(h, hipGetLastError());
        if (has_next && lookahead) DCP_TRY(main_after_side(h));   // (also frees the ext scratch for (4))
        {   // (4) D_new[blk] = E . P
            GemmArgs<T> a;
            a.A = w.E; a.lda = nb; a.B = P; a.ldb = F; a.M = nb; a.N = F; a.K = nb;
            a.A_rows = w.rows_E; a.lda_rows = nb;
            a.tile = TILE_SMALL_DEEP;
            a.ext_ws = w.ext;
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a, EpiStore<T>{Dnew + (long)k0 * F, (long)F})));
        }
        if (has_next) {   // (5) P_next += Aprev[next block] . D_new[blk]
            GemmArgs<T> a;
            a.A = w.Aprev + (long)k1 * kAtomBlkMax; a.lda = kAtomBlkMax; a.B = Dnew + (long)k0 * F; a.ldb = F;
            a.M = nb1; a.N = F; a.K = nb;
            if (w.rows_prev) { a.A_rows = w.rows_prev + 2L * k1 * kAtomBlkMax; a.lda_rows = kAtomBlkMax; }
            a.tile = TILE_SMALL_DEEP;
            a.ext_ws = w.ext;
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a, EpiAddTo<T>{Pnext, (long)F})));
        }
    }
    return DCP_OK;
}

}  // namespace dcp
