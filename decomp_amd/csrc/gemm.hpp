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
enum TileSel { TILE_AUTO = 0, TILE_LARGE = 1, TILE_SMALL = 2, TILE_HUGE = 3 };

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
    bool split_planned = false;  // set by plan_splits: tile tiers may count on split-K
};

// MFMA tile tiers (BM, BN, BK, WM, WN, min waves/SIMD).  Measured on MI355X (tools/gemm_sweep.py,
// Y.D^T at 65536x4096x256): 256x256 / 16 waves 139 TF, 128x128 / 4 waves 110 TF -- the big tile
// halves the operand traffic per flop and one workgroup per CU is exactly one round.  The
// reduction-over-samples form (TN) is insensitive to the tile (131 TF either way) and keeps
// 128x128, whose 68 tiles x 15 splits fill one round of 1024 resident workgroups.
typedef TileCfg<256, 256, 16, 64, 64, 1> CfgHuge;      // 16 waves, 64 KiB LDS
typedef TileCfg<128, 128, 16, 64, 64, 2> CfgLarge;     // 4 waves, 32 KiB LDS -> 4 WG/CU
typedef TileCfg<64, 64, 16, 32, 32, 2> CfgSmall;
typedef TileCfg<32, 128, 32, 32, 32, 2> CfgFlat;       // <= 32 output rows (atom-block residuals)

enum Tier { TIER_SMALL = 0, TIER_LARGE = 1, TIER_HUGE = 2, TIER_FLAT = 3 };

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

inline void tier_dims(int tier, int& bm, int& bn) {
    if (tier == TIER_HUGE) { bm = CfgHuge::BM; bn = CfgHuge::BN; }
    else if (tier == TIER_LARGE) { bm = CfgLarge::BM; bn = CfgLarge::BN; }
    else if (tier == TIER_FLAT) { bm = CfgFlat::BM; bn = CfgFlat::BN; }
    else { bm = 64; bn = 64; }
}

// The tile tier the float path uses for this problem (split-K planning needs it before the
// launch): the largest tile that still gives the chip enough workgroups, counting the splits
// a deep reduction allows.
template <int FORM>
inline int pick_tier(int M, int N, int K, int tile_sel, bool will_split) {
    if (tile_sel == TILE_SMALL) return TIER_SMALL;
    if (tile_sel == TILE_LARGE) return TIER_LARGE;
    if (tile_sel == TILE_HUGE) return FORM == FORM_TN ? TIER_LARGE : TIER_HUGE;
    long splits = will_split ? K / 512 : 1;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    if (FORM != FORM_TN) {
        const long wh = (long)ceil_div(M, CfgHuge::BM) * ceil_div(N, CfgHuge::BN);
        if (M >= 256 && N >= 256 && wh * splits >= 192) return TIER_HUGE;
    }
    // un-split products want at least two 4-wave workgroups per CU to hide latency
    const long wl = (long)ceil_div(M, CfgLarge::BM) * ceil_div(N, CfgLarge::BN);
    if (wl * splits >= (will_split ? 256 : 512)) return TIER_LARGE;
    return TIER_SMALL;
}

// Choose split-K so that the grid reaches ~target workgroups, each split a multiple
// of 16 deep (the MFMA K block).  Returns the number of splits; sets a.klen.
template <int FORM, class T>
inline int plan_splits(GemmArgs<T>& a, int target_wgs, int max_splits) {
    int bm = 64, bn = 64;
    a.split_planned = true;
    if (std::is_same<T, float>::value) tier_dims(pick_tier<FORM>(a.M, a.N, a.K, a.tile, true), bm, bn);
    const int n1 = a.B2 != nullptr ? a.n_b1 : a.N;
    const long tiles = (long)ceil_div(a.M, bm) * (ceil_div(n1, bn) + ceil_div(a.N - n1, bn));
    const long kblocks = ceil_div(a.K > 0 ? a.K : 1, 16);
    // splits allowed by the reduction depth: keep every split at least 512 deep (32 K blocks)
    // so that tile prologue / slab write-out stay small against the MFMA work
    long smax = kblocks / 32;
    if (smax < 1) smax = 1;
    if (smax > max_splits) smax = max_splits;
    long s = tiles > 0 ? target_wgs / tiles : 1;  // floor: stay within `target` resident slots
    if (s > smax) s = smax;
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
        const int tier = pick_tier<FORM>(a.M, a.N, a.K, a.tile, a.split_planned);
        if (tier == TIER_SMALL) return launch_gemm_mfma<CfgSmall, AL, BL, Epi>(stream, p, epi);
        if constexpr (FORM != FORM_TN) {
            if (tier == TIER_HUGE) return launch_gemm_mfma<CfgHuge, AL, BL, Epi>(stream, p, epi);
        }
        return launch_gemm_mfma<CfgLarge, AL, BL, Epi>(stream, p, epi);
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
