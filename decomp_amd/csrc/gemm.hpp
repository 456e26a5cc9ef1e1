// One GEMM front end for every dtype: float goes to the fp32 MFMA core
// (gemm_mfma_f32.hpp), double / complex to the generic LDS-tiled core
// (gemm_generic.hpp).  Epilogue functors (below) fuse the per-element update rules
// of the solvers into the GEMM that produces their last operand.
#pragma once
#include <type_traits>

#include "gemm_generic.hpp"
#include "gemm_mfma_f32.hpp"
#include "scalar.hpp"

namespace dcp {

// Storage forms (all row-major, as the reference's NumPy arrays):
//   NT: C[M,N] = A[M,K] . B[N,K]^T(H)      NN: C[M,N] = A[M,K] . B[K,N]
//   TN: C[M,N] = A[K,M]^T(H) . B[K,N]
enum GemmForm { FORM_NT = 0, FORM_NN = 1, FORM_TN = 2 };
enum TileSel { TILE_AUTO = 0, TILE_LARGE = 1, TILE_SMALL = 2 };

template <class T>
struct GemmArgs {
    const T* A = nullptr;
    long lda = 0;
    const T* B = nullptr;
    long ldb = 0;
    const T* B2 = nullptr;  // optional second column segment of B (columns >= n_b1)
    long ldb2 = 0;
    int n_b1 = 0;
    int M = 0, N = 0, K = 0;
    int ksplits = 1;  // >1: split the reduction; the epilogue sees the split index
    int klen = 0;     // reduction length per split (set by plan_splits)
    bool conjA = false, conjB = false;  // complex only
    int tile = TILE_AUTO;
};

// MFMA tile configurations (BM, BN, BK, WM, WN, min waves/SIMD)
typedef TileCfg<128, 128, 16, 64, 64, 2> CfgLargeNT;   // 64 acc regs/lane -> up to 4 WG/CU
typedef TileCfg<128, 128, 16, 64, 64, 2> CfgLargeNN;
typedef TileCfg<128, 128, 16, 64, 64, 2> CfgLargeTN;
typedef TileCfg<64, 64, 16, 32, 32, 2> CfgSmall;

template <int FORM> struct LargeCfgOf;
template <> struct LargeCfgOf<FORM_NT> { typedef CfgLargeNT type; };
template <> struct LargeCfgOf<FORM_NN> { typedef CfgLargeNN type; };
template <> struct LargeCfgOf<FORM_TN> { typedef CfgLargeTN type; };

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Tile footprint the float path will use for this problem (needed to plan split-K
// before the launch).  Large tiles when they alone give the chip enough workgroups.
template <int FORM>
inline bool use_small_tile(int M, int N, int K, int tile_sel) {
    if (tile_sel == TILE_LARGE) return false;
    if (tile_sel == TILE_SMALL) return true;
    typedef typename LargeCfgOf<FORM>::type L;
    const long wgs_large = (long)ceil_div(M, L::BM) * ceil_div(N, L::BN);
    // a deep reduction can still fill the chip with large tiles through split-K
    long splits = K / 512;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    return wgs_large * splits < 256;
}

// Choose split-K so that the grid reaches ~target workgroups, each split a multiple
// of 16 deep (the MFMA K block).  Returns the number of splits; sets a.klen.
template <int FORM, class T>
inline int plan_splits(GemmArgs<T>& a, int target_wgs, int max_splits) {
    int bm = 64, bn = 64;
    if (std::is_same<T, float>::value && !use_small_tile<FORM>(a.M, a.N, a.K, a.tile)) {
        typedef typename LargeCfgOf<FORM>::type L;
        bm = L::BM;
        bn = L::BN;
    }
    const int n1 = a.B2 != nullptr ? a.n_b1 : a.N;
    const long tiles = (long)ceil_div(a.M, bm) * (ceil_div(n1, bn) + ceil_div(a.N - n1, bn));
    long s = tiles > 0 ? target_wgs / tiles : 1;  // floor: stay within `target` resident slots
    if (s > max_splits) s = max_splits;
    const long kblocks = ceil_div(a.K > 0 ? a.K : 1, 16);
    if (s > kblocks) s = kblocks;
    if (s < 1) s = 1;
    const long blocks_per_split = (kblocks + s - 1) / s;
    a.klen = (int)(blocks_per_split * 16);
    a.ksplits = ceil_div(a.K > 0 ? a.K : 1, a.klen);
    return a.ksplits;
}

template <int FORM, class T, class Epi>
inline hipError_t gemm(hipStream_t stream, const GemmArgs<T>& a, const Epi& epi) {
    if constexpr (std::is_same<T, float>::value) {
        GemmProblem p;
        p.A = a.A; p.lda = a.lda; p.B = a.B; p.ldb = a.ldb;
        p.B2 = a.B2; p.ldb2 = a.ldb2; p.n_b1 = a.n_b1;
        p.M = a.M; p.N = a.N; p.K = a.K;
        p.ksplits = a.ksplits; p.klen = a.klen;
        p.tiles_m = p.tiles_n = 0;
        // NT walks n tiles first (they share the A row panel); TN/NN walk m first.
        p.mt_fast = (FORM == FORM_NT) ? 0 : 1;
        constexpr int AL = (FORM == FORM_TN) ? XMAJOR : KMAJOR;
        constexpr int BL = (FORM == FORM_NT) ? KMAJOR : XMAJOR;
        if (use_small_tile<FORM>(a.M, a.N, a.K, a.tile))
            return launch_gemm_mfma<CfgSmall, AL, BL, Epi>(stream, p, epi);
        return launch_gemm_mfma<typename LargeCfgOf<FORM>::type, AL, BL, Epi>(stream, p, epi);
    } else {
        GenericProblem<T> p;
        p.A = a.A; p.B = a.B; p.B2 = a.B2; p.n_b1 = a.n_b1;
        if (FORM == FORM_TN) { p.sAm = 1; p.sAk = a.lda; } else { p.sAm = a.lda; p.sAk = 1; }
        if (FORM == FORM_NT) { p.sBk = 1; p.sBn = a.ldb; p.sB2k = 1; p.sB2n = a.ldb2; }
        else { p.sBk = a.ldb; p.sBn = 1; p.sB2k = a.ldb2; p.sB2n = 1; }
        p.M = a.M; p.N = a.N; p.K = a.K;
        p.ksplits = a.ksplits; p.klen = a.klen;
        p.tiles_m = p.tiles_n = 0;
        if constexpr (scalar_traits<T>::is_complex)
            return launch_gemm_generic<T, Epi>(stream, p, a.conjA, a.conjB, epi);
        else
            return launch_gemm_generic<T, Epi>(stream, p, false, false, epi);
    }
}

// ------------------------------------------------------------------ epilogues ----
// All are called as epi(row, col, acc, split).

template <class T>
struct EpiStore {  // C[row, col] = acc
    T* C;
    long ldc;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        C[(long)r * ldc + c] = v;
    }
};

template <class T>
struct EpiSlab {  // split-K partial: slab[split][row, col] = acc
    T* slab;
    long ldc;
    long slab_stride;
    __device__ __forceinline__ void operator()(int r, int c, T v, int s) const {
        slab[(long)s * slab_stride + (long)r * ldc + c] = v;
    }
};

// out = cur * max(acc, 0) / max(den, 1e-15): acc is the POSITIVE gradient part
// (grads.py:84 with grad_pos produced by this GEMM).  den: full matrix (ld_den > 0)
// or one value per column (ld_den == 0), e.g. KL's colsum(D).
template <class T>
struct EpiMuNum {
    const T* cur;
    long ld_cur;
    const T* den;
    long ld_den;
    T* out;
    long ld_out;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        const T d = den[(long)r * ld_den + c];
        out[(long)r * ld_out + c] = cur[(long)r * ld_cur + c] * (v > T(0) ? v : T(0)) /
                                    (d > T(1.0e-15) ? d : T(1.0e-15));
    }
};

// out = cur * max(num, 0) / max(acc, 1e-15): acc is the NEGATIVE gradient part.
template <class T>
struct EpiMuDen {
    const T* cur;
    long ld_cur;
    const T* num;
    long ld_num;
    T* out;
    long ld_out;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        const T nu = num[(long)r * ld_num + c];
        out[(long)r * ld_out + c] = cur[(long)r * ld_cur + c] * (nu > T(0) ? nu : T(0)) /
                                    (v > T(1.0e-15) ? v : T(1.0e-15));
    }
};

// out = acc * mask  (f = (x.D) o M, grads.py:113,123)
template <class T>
struct EpiMulMask {
    const real_t<T>* mask;
    long ld_mask;
    T* out;
    long ld_out;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        out[(long)r * ld_out + c] = scale(v, mask[(long)r * ld_mask + c]);
    }
};

// KL ratio: out = (y [* mask]) / (acc + 1e-15)   (grads.py:145-149,154-158)
template <class T>
struct EpiKlRatio {
    const T* y;
    long ld_y;
    const T* mask;  // nullable
    long ld_mask;
    T* out;
    long ld_out;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        T yy = y[(long)r * ld_y + c];
        if (mask != nullptr) yy = yy * mask[(long)r * ld_mask + c];
        out[(long)r * ld_out + c] = yy / (v + T(1.0e-15));
    }
};

// Residual: out = (y - acc) [* mask]   (parity metric; reduced by sumsq_partial_kernel)
template <class T>
struct EpiResidual {
    const T* y;
    long ld_y;
    const real_t<T>* mask;  // nullable
    long ld_mask;
    T* out;
    long ld_out;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        T d = sub(y[(long)r * ld_y + c], v);
        if (mask != nullptr) d = scale(d, mask[(long)r * ld_mask + c]);
        out[(long)r * ld_out + c] = d;
    }
};

}  // namespace dcp
