// One GEMM front end for every dtype: float and complex64 go to the fp32 MFMA core
// (gemm_mfma_f32.hpp), double and complex128 to the fp64 MFMA core (gemm_mfma_f64.hpp) when the
// output is at least 128 x 128; what is left (thin fp64 / complex128 outputs, complex products
// without an extended-operand scratch) runs on the generic LDS-tiled VALU core (gemm_generic.hpp).  Epilogue functors (below) fuse the per-element update rules
// of the solvers into the GEMM that produces their last operand.
#pragma once
#include <type_traits>

#include "gemm_generic.hpp"
#include "gemm_mfma_f32.hpp"
#include "gemm_mfma_f64.hpp"
#include "scalar.hpp"

namespace dcp {

// Storage forms (all row-major, as the reference's NumPy arrays):
//   NT: C[M,N] = A[M,K] . B[N,K]^T(H)      NN: C[M,N] = A[M,K] . B[K,N]
//   TN: C[M,N] = A[K,M]^T(H) . B[K,N]
enum GemmForm { FORM_NT = 0, FORM_NN = 1, FORM_TN = 2 };
enum TileSel { TILE_AUTO = 0, TILE_LARGE = 1, TILE_SMALL = 2, TILE_HUGE = 3, TILE_SMALL_DEEP = 4, TILE_MID = 5 };

template <class T>
struct GemmArgs {
    const T* A = nullptr;
    long lda = 0;
    const T* B = nullptr;
    long ldb = 0;
    const T* B2 = nullptr;  // optional second column segment of B (columns >= n_b1)
    long ldb2 = 0;
    int n_b1 = 0;
    const T* A2 = nullptr;  // optional second ROW segment of A (rows >= m_a1); float32 MFMA core only
    long lda2 = 0;
    int m_a1 = 0;
    int M = 0, N = 0, K = 0;
    int ksplits = 1;  // >1: split the reduction; the epilogue sees the split index
    int klen = 0;     // reduction length per split (set by plan_splits)
    bool conjA = false, conjB = false;  // complex only
    int tile = TILE_AUTO;
    bool split_planned = false;  // set by plan_splits: tile tiers may count on split-K
    // complex only: scratch for the real "extended" image of B (4 * rows(B) * cols(B) reals).
    // With it NT / NN products run on the fp32 / fp64 MFMA core; TN needs none.  Null -> generic core.
    real_t<T>* ext_ws = nullptr;
    // complex only: ext_ws ALREADY holds the extended image of this B (same B, same form family) from an
    // earlier product of the caller -- the ten ISTA iterations of a solve multiply by the same A A^H
    bool ext_ready = false;
    // complex64 NN only: the planar-rows image of A ([2M, K] reals, leading dim lda_rows) when the caller
    // already holds it (the atom sweep prepares its block matrices that way once per sweep)
    const real_t<T>* A_rows = nullptr;
    long lda_rows = 0;
};

// complex64 products on the fp32 MFMA core --------------------------------------------------
// ext(B) for a complex [R, C] matrix: real [2R, 2C], row 2r = (re, im) pairs (B's own memory
// image), row 2r+1 = (-im, re).  With the interleaved real view A2 of the other operand:
//   NT, C = A B^H : C_real[m][2n], [2n+1] = A2[m, :] . ext(B)[2n], [2n+1]   (K' = 2K)
//   NN, C = A B   : C_real[m][2n], [2n+1] = sum_kk A2[m][kk] ext(B)[kk][2n], [2n+1]
//   TN, C = A^H B : real views of both operands, 2x2 blocks combined in the epilogue (mode 2)
// i.e. the 4 real multiplies of every complex multiply, 8MNK flops, no operand copies except
// the small ext(B).
template <class R>
__global__ void __launch_bounds__(256) cplx_ext_kernel(const cx<R>* __restrict__ B, long rows, long cols,
                                                       long ld, R* __restrict__ out) {
    const long n = rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const long r = i / cols, c = i - r * cols;
        const cx<R> v = B[r * ld + c];
        R* o0 = out + (2 * r) * (2 * cols) + 2 * c;
        R* o1 = o0 + 2 * cols;
        o0[0] = v.re; o0[1] = v.im;
        o1[0] = -v.im; o1[1] = v.re;
    }
}

// Planar rows of a complex [R, C] matrix: real [2R, C], row 2r = Re A[r, :], row 2r+1 = Im A[r, :].
// With it a product A B whose LEFT operand is the small one needs no image of B at all:
//   NN, C = A B : G = rows(A) . B_real (B's own memory as [K, 2N]); the 2x2 block (rr ri; ir ii) of G gives
//   re = rr - ii, im = ri + ir in the epilogue (mode 3)
// (atom-block products [32 x K].[K x F]: the image of the F-long operand was 2/3 of their time).
template <class R>
__global__ void __launch_bounds__(256) cplx_rows_kernel(const cx<R>* __restrict__ A, long rows, long cols,
                                                        long ld, R* __restrict__ out) {
    const long n = rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const long r = i / cols, c = i - r * cols;
        const cx<R> v = A[r * ld + c];
        out[(2 * r) * cols + c] = v.re;
        out[(2 * r + 1) * cols + c] = v.im;
    }
}

// A complex64 functor may carry `static constexpr bool kCVec2 = true` with
//     bool cvec_ok() const                                   (host: arrays 16-byte aligned, leading dims even)
//     void cvec2(int row, int col0, f32x4 v, int split) const   (v = re, im of columns col0 and col0 + 1)
// In mode 1 the real accumulator columns 2n / 2n+1 ARE (re, im), so the core's 16-byte epilogue (epi_vec4)
// hands a lane two complex outputs of one row: all lanes active, 16 bytes per load / store, instead of
// 8 bytes on the even lanes.
template <class E, class = void>
struct epi_cvec2 { static constexpr bool value = false; };
template <class E>
struct epi_cvec2<E, decltype((void)E::kCVec2)> { static constexpr bool value = E::kCVec2; };

template <class E, class R = float>
struct CplxColEpi {   // kernel epilogue mode 1
    static constexpr int kMode = 1;
    static constexpr bool kVec4 = std::is_same<R, float>::value && epi_cvec2<E>::value;
    E e;
    __device__ __forceinline__ void pair(int r, int c, R re, R im, int s) const {
        e(r, c, cx<R>{re, im}, s);
    }
    __device__ __forceinline__ void operator()(int, int, R, int) const {}
    bool vec_ok() const {
        if constexpr (kVec4) return e.cvec_ok();
        else return false;
    }
    __device__ __forceinline__ void vec4(int r, int c0, f32x4 v, int s) const {
        if constexpr (kVec4) e.cvec2(r, c0 >> 1, v, s);
    }
};
template <class E, class R = float>
struct CplxTnEpi {    // kernel epilogue mode 2
    static constexpr int kMode = 2;
    E e;
    __device__ __forceinline__ void pair(int r, int c, R re, R im, int s) const {
        e(r, c, cx<R>{re, im}, s);
    }
    __device__ __forceinline__ void operator()(int, int, R, int) const {}
};

template <class E, class R = float>
struct CplxNnEpi {    // kernel epilogue mode 3
    static constexpr int kMode = 3;
    E e;
    __device__ __forceinline__ void pair(int r, int c, R re, R im, int s) const {
        e(r, c, cx<R>{re, im}, s);
    }
    __device__ __forceinline__ void operator()(int, int, R, int) const {}
};

// complex64 NN products whose left operand is the smaller one take the planar-rows form
template <int FORM>
inline bool cplx_planar_a(int M, int N, bool conjA, bool conjB, const void* ext_ws) {
    return FORM == FORM_NN && !conjA && !conjB && ext_ws != nullptr && M < N;
}

template <int FORM>
inline bool cplx_on_mfma(bool conjA, bool conjB, const void* ext_ws) {
    if (FORM == FORM_TN) return conjA && !conjB;
    if (FORM == FORM_NT) return conjB && !conjA && ext_ws != nullptr;
    return !conjA && !conjB && ext_ws != nullptr;
}

// MFMA tile tiers (BM, BN, BK, WM, WN, min waves/SIMD).  Measured on MI355X (tools/gemm_sweep.py,
// Y.D^T at 65536x4096x256): 256x256 / 16 waves 139 TF, 128x128 / 4 waves 110 TF -- the big tile
// halves the operand traffic per flop and one workgroup per CU is exactly one round.  The
// reduction-over-samples form (TN) is insensitive to the tile (131 TF either way) and keeps
// 128x128, whose 68 tiles x 15 splits fill one round of 1024 resident workgroups.
typedef TileCfg<256, 256, 16, 64, 64, 1> CfgHuge;      // 16 waves, 64 KiB LDS
typedef TileCfg<128, 128, 16, 64, 64, 2> CfgLarge;     // 4 waves, 32 KiB LDS -> 4 WG/CU
typedef TileCfg<128, 128, 64, 32, 64, 2> CfgMid;       // 8 waves of 32x64, 64-deep K blocks (128 KiB LDS): mid-size
                                                      // outputs (128..511 tiles, at most one per CU), e.g. y.A^H of
                                                      // an 8192-row minibatch, 8192 x 512 x 4096: 0.292 ms with 16-deep
                                                      // blocks, 0.275 ms with 64-deep ones (vendor BLAS: 0.263 ms)
typedef TileCfg<64, 64, 16, 32, 32, 2> CfgSmall;
typedef TileCfg<32, 128, 32, 32, 32, 2> CfgFlat;       // <= 32 output rows (x^T [Y|x] with <= 32 atoms)
typedef TileCfg<128, 32, 32, 32, 32, 2> CfgTall;       // <= 32 output columns (Y.D^T, x.G with <= 32 atoms)
typedef TileCfg<256, 256, 32, 64, 64, 1> CfgHugeDeep;   // 16 waves, 32-deep K blocks (128 KiB LDS): x^T [y|x] with >= 512
                                                      // atoms (dictionary step, 512 x 4608 x 8192, 7 splits: 0.362 -> 0.328 ms)
typedef TileCfg<128, 128, 16, 64, 64, 2, 3> CfgLargeX;       // the pair schedule (PIPE = 3) for row-contiguous panels
typedef TileCfg<256, 256, 32, 64, 64, 1, 3> CfgHugeDeepX;
typedef TileCfg<64, 64, 64, 32, 32, 1> CfgSmallDeep;   // 64 KiB LDS: latency-bound products on few CUs (64-row
                                                      // atom-block GEMMs): 4x fewer, 4x larger K blocks in flight

enum Tier { TIER_SMALL = 0, TIER_LARGE = 1, TIER_HUGE = 2, TIER_FLAT = 3, TIER_MID = 4, TIER_SMALL_DEEP = 5, TIER_TALL = 6, TIER_HUGE_DEEP = 7 };

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

inline void tier_dims(int tier, int& bm, int& bn) {
    if (tier == TIER_HUGE || tier == TIER_HUGE_DEEP) { bm = CfgHuge::BM; bn = CfgHuge::BN; }
    else if (tier == TIER_LARGE || tier == TIER_MID) { bm = CfgLarge::BM; bn = CfgLarge::BN; }
    else if (tier == TIER_FLAT) { bm = CfgFlat::BM; bn = CfgFlat::BN; }
    else if (tier == TIER_TALL) { bm = CfgTall::BM; bn = CfgTall::BN; }
    else { bm = 64; bn = 64; }
}

// The tile tier the float path uses for this problem (split-K planning needs it before the
// launch): the largest tile that still gives the chip enough workgroups, counting the splits
// a deep reduction allows.
template <int FORM>
inline int pick_tier(int M, int N, int K, int tile_sel, bool will_split, bool real32 = false) {
    if (tile_sel == TILE_SMALL) return TIER_SMALL;
    if (tile_sel == TILE_SMALL_DEEP) return TIER_SMALL_DEEP;
    if (tile_sel == TILE_LARGE) return TIER_LARGE;
    if (tile_sel == TILE_MID) return (FORM != FORM_TN && M >= 128 && N >= 128) ? TIER_MID : TIER_SMALL;
    if (tile_sel == TILE_HUGE) return FORM == FORM_TN ? TIER_LARGE : TIER_HUGE;
    // <= 64 output rows or columns (<= 64 atoms): a 128-wide tile would idle half of every MFMA and run
    // bounds-checked (Y.D^T 65536 x 64 x 4096: 0.60 ms on 128x128, 0.34 ms on 64x64; x^T Y 64 x 4160 x 65536: 1.07 -> 0.39 ms)
    // <= 32 (the usual NMF ranks): 32-wide tiles; these products are HBM bound (Y is read once per product:
    // Y.D^T 65536 x 32 x 4096 0.40 ms on 64x64, 0.22 ms = 4.9 TB/s on 128x32; x^T Y 32 x 4128 x 65536 0.53 -> 0.32 ms)
    if (M <= 32 && N > 32) return TIER_FLAT;
    if (N <= 32 && M > 32) return TIER_TALL;
    if (M <= 64 || N <= 64) return TIER_SMALL;
    long splits = will_split ? K / 512 : 1;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    // reduction over samples with a tall output (>= 512 atoms), split over the samples: the big tile with deep
    // K blocks (float32 problems only: the complex paths launch 64- or 128-wide tiles for this form)
    if (FORM == FORM_TN && real32 && will_split && M >= 512 && N >= 512 && tile_sel == TILE_AUTO) return TIER_HUGE_DEEP;
    if (FORM != FORM_TN) {
        const long wh = (long)ceil_div(M, CfgHuge::BM) * ceil_div(N, CfgHuge::BN);
        // (N in (128, 256), (384, 512), ...: the 256-wide tile covers no more columns than 128-wide tiles would:
        //  Y.D^T 65536 x 252 x 4096 1.117 -> 1.064 ms)
        const bool cols_ok = N >= 256 || ceil_div(N, CfgHuge::BN) * CfgHuge::BN == ceil_div(N, CfgLarge::BN) * CfgLarge::BN;
        if (M >= 256 && cols_ok && N > 128 && wh * splits >= 192) return TIER_HUGE;
    }
    // un-split products want at least two 4-wave workgroups per CU to hide latency
    const long wl = (long)ceil_div(M, CfgLarge::BM) * ceil_div(N, CfgLarge::BN);
    if (wl * splits >= (will_split ? 256 : 512)) return TIER_LARGE;
    // one 128x128 tile per CU or fewer: 8 waves per tile instead of 4 keep two waves on every SIMD
    // (deep reductions only: measured on y.A^H, K = 4096, +8 %; for K = 256..512 products -- x.G, the
    //  ISTA step -- the 64x64 tile with its 4x more workgroups is as fast or faster)
    if (FORM != FORM_TN && M >= 128 && N >= 128 && K >= 1024 && wl * splits >= 128) return TIER_MID;
    return TIER_SMALL;
}

// float64 (the reference's default dtype) and complex128 (on real-extended operands) tiles on the fp64 MFMA
// core.  Narrow outputs -- the usual NMF
// ranks of 8 .. 64 atoms -- get 32- and 64-wide tiles; before, anything under 128 fell to the generic VALU core
// (MU iteration 16384 x 4096, float64: k = 100 2.06 ms, k = 64 0.92, k = 32 0.84, k = 8 0.79 ms).
enum F64Tier { F64_GENERIC = 0, F64_128 = 1, F64_TALL64 = 2, F64_TALL32 = 3, F64_FLAT64 = 4, F64_FLAT32 = 5, F64_SQ64 = 6 };
typedef F64Cfg<64, 64, 32, 32, 1> F64Sq64;      // 4 waves: block Gram matrices of the atom sweep (deep reduction, split)
typedef F64Cfg<128, 64, 32, 32, 1> F64Tall64;   // 8 waves of 32 x 32
typedef F64Cfg<128, 32, 32, 32, 2> F64Tall32;   // 4 waves
typedef F64Cfg<64, 128, 32, 32, 1> F64Flat64;   // 8 waves
typedef F64Cfg<32, 128, 32, 32, 2> F64Flat32;   // 4 waves

inline int f64_tier(int M, int N, int tile_sel, int K = 1 << 30) {
    // (TILE_SMALL_DEEP -- the atom sweep's 64-row block products -- is a float32 tile choice; in double precision
    //  those products take the 64 x 128 tile here: the generic core needed 113 us for the 64 x 4096 x 512 one)
    if (tile_sel == TILE_SMALL) return F64_GENERIC;
    // the caller split a deep reduction over a small square output itself (Gram matrix D D^T, 256 x 256 x 4096 in 16
    // splits): 64 x 64 tiles give 256 workgroups where 128 x 128 tiles give 64 (37 -> 13 us)
    if (tile_sel == TILE_SMALL_DEEP && M > 64 && N > 64 && M <= 1024 && N <= 1024) return F64_SQ64;
    if (M > 64 && N > 64) {
        // few 128 x 128 tiles (the D-side products of an MU step: 256 x 4096 -> 64 tiles on 256 CUs): 64 x 64 tiles
        // put four times as many workgroups on the chip; these products are latency bound
        const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        if (t128 < 96 && K <= 1024) return F64_SQ64;   // (shallow reductions only)
        // one 128-wide strip of tiles that cannot fill the chip (128 atoms, 16384 rows: 128 tiles): half-width tiles
        if (t128 < 192 && N <= 128 && M >= 256) return F64_TALL64;
        if (t128 < 192 && M <= 128 && N >= 256) return F64_FLAT64;
        return F64_128;
    }
    if (N <= 32 && M >= 128) return F64_TALL32;
    if (N <= 64 && M >= 128) return F64_TALL64;
    if (M <= 32 && N >= 128) return F64_FLAT32;
    if (M <= 64 && N >= 128) return F64_FLAT64;
    if (M > 32 && N > 32) return F64_SQ64;   // (33..64) x (33..64 or 65..127 handled above): small square outputs
    return F64_GENERIC;
}
inline bool f64_tier_dims(int tier, int& bm, int& bn) {
    switch (tier) {
        case F64_128: bm = 128; bn = 128; return true;
        case F64_TALL64: bm = 128; bn = 64; return true;
        case F64_TALL32: bm = 128; bn = 32; return true;
        case F64_FLAT64: bm = 64; bn = 128; return true;
        case F64_FLAT32: bm = 32; bn = 128; return true;
        case F64_SQ64: bm = 64; bn = 64; return true;
        default: return false;
    }
}

template <int AL, int BL, class Epi>
inline hipError_t launch_f64_tier(int tier, hipStream_t stream, const GemmProblemD& p, const Epi& epi) {
    if (tier == F64_TALL64) return launch_gemm_mfma_f64_cfg<F64Tall64, AL, BL, Epi>(stream, p, epi);
    if (tier == F64_TALL32) return launch_gemm_mfma_f64_cfg<F64Tall32, AL, BL, Epi>(stream, p, epi);
    if (tier == F64_FLAT64) return launch_gemm_mfma_f64_cfg<F64Flat64, AL, BL, Epi>(stream, p, epi);
    if (tier == F64_FLAT32) return launch_gemm_mfma_f64_cfg<F64Flat32, AL, BL, Epi>(stream, p, epi);
    if (tier == F64_SQ64) return launch_gemm_mfma_f64_cfg<F64Sq64, AL, BL, Epi>(stream, p, epi);
    return launch_gemm_mfma_f64<AL, BL, Epi>(stream, p, epi);
}

// Choose split-K so that the grid reaches ~target workgroups, each split a multiple
// of 16 deep (the MFMA K block).  Returns the number of splits; sets a.klen.
template <int FORM, class T>
inline int plan_splits(GemmArgs<T>& a, int target_wgs, int max_splits, int min_blocks = 32) {
    int bm = 64, bn = 64;
    a.split_planned = true;
    // problem dims as the MFMA core sees them (complex64: real-extended)
    int Mx = a.M, Nx = a.N, Kx = a.K, n1 = a.B2 != nullptr ? a.n_b1 : a.N;
    bool mfma = std::is_same<T, float>::value;
    if (std::is_same<T, c64>::value && cplx_on_mfma<FORM>(a.conjA, a.conjB, a.ext_ws)) {
        mfma = true;
        Nx *= 2; n1 *= 2;
        if (FORM == FORM_TN || cplx_planar_a<FORM>(a.M, a.N, a.conjA, a.conjB, a.ext_ws)) Mx *= 2;
        else Kx *= 2;
    }
    int kunit = 1;   // K blocks of 16 per K block of the tile that will run (splits are whole tile blocks)
    if (mfma) {
        const int tier = pick_tier<FORM>(Mx, Nx, Kx, a.tile, true, std::is_same<T, float>::value);
        tier_dims(tier, bm, bn);
        if (std::is_same<T, float>::value) {
            if (tier == TIER_HUGE_DEEP) kunit = CfgHugeDeep::BK / 16;
            if (tier == TIER_MID) kunit = CfgMid::BK / 16;
        }
    }
    if (mfma && FORM == FORM_TN && bm == CfgHuge::BM && target_wgs > 256) target_wgs = 256;   // one 16-wave tile per CU
    if (std::is_same<T, double>::value) (void)f64_tier_dims(f64_tier(a.M, a.N, a.tile, a.K), bm, bn);
    if (std::is_same<T, c128>::value && cplx_on_mfma<FORM>(a.conjA, a.conjB, a.ext_ws)) {
        const int Me = (FORM == FORM_TN) ? 2 * a.M : a.M;
        int tbm = 0, tbn = 0;
        const bool planar = cplx_planar_a<FORM>(a.M, a.N, a.conjA, a.conjB, a.ext_ws) &&
                            f64_tier(2 * a.M, 2 * a.N, a.tile) != F64_GENERIC;
        if (f64_tier_dims(f64_tier(planar ? 2 * a.M : Me, 2 * a.N, a.tile), tbm, tbn)) {   // fp64 MFMA core
            Mx = planar ? 2 * a.M : Me; Nx = 2 * a.N; n1 *= 2;
            if (FORM != FORM_TN && !planar) Kx *= 2;
            bm = tbm; bn = tbn;
        }
    }
    const long tiles = (long)ceil_div(Mx, bm) * (ceil_div(n1, bn) + ceil_div(Nx - n1, bn));
    const long kblocks = ceil_div(Kx > 0 ? Kx : 1, 16);   // (real-extended depth for complex64)
    // splits allowed by the reduction depth: keep every split at least 512 deep (32 K blocks)
    // so that tile prologue / slab write-out stay small against the MFMA work
    long smax = kblocks / min_blocks;
    if (smax < 1) smax = 1;
    if (smax > max_splits) smax = max_splits;
    long s = tiles > 0 ? target_wgs / tiles : 1;  // floor: stay within `target` resident slots
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    if (FORM == FORM_TN && mfma && bm == CfgLarge::BM && s == 1 && tiles > target_wgs) {
        // A grid just over a whole number of rounds of resident 128x128 workgroups (complex64 minibatch
        // statistics: 1088 tiles on 1024 slots) leaves the chip nearly empty for a whole tile's time: shorter
        // workgroups shrink that tail.  Cost model: rounds / s + 1.5 % per extra set of partial slabs; every
        // split at least 1024 deep.  Measured 1024 x 17408 x 8192: 1 split 2.63 ms, 2: 2.42, 3: 2.32, 4: 2.29, 5: 2.40.
        double best = 1e30;
        long best_s = 1;
        for (long c = 1; c <= 4 && c <= kblocks / 64 && c <= max_splits; ++c) {
            const double rounds = (double)((tiles * c + target_wgs - 1) / target_wgs);
            const double cost = rounds / (double)c + 0.015 * (double)c;
            if (cost < best - 1e-9) { best = cost; best_s = c; }
        }
        s = best_s;
    }
    long blocks_per_split = (kblocks + s - 1) / s;
    blocks_per_split = ((blocks_per_split + kunit - 1) / kunit) * kunit;
    a.klen = (int)(blocks_per_split * 16);   // in units of the core's reduction index (see gemm())
    a.ksplits = ceil_div(Kx > 0 ? Kx : 1, a.klen);
    return a.ksplits;
}

// A . B^H with a deep reduction whose 256x256 tiles alone cannot fill the chip (complex64 minibatch y.A^H:
// 8192 x 1024 x 16384 as the real core sees it -> 128 tiles): S splits on the big tile instead of the un-split
// 128x128 tile (measured 2.27 -> 1.99 ms including the ordered slab sum).  Every split stays >= 8192 deep: at
// 4096 (float32 minibatch, 64 tiles) the two plans tie.  Returns S and plans `a` for it; 1 leaves `a` untouched.
template <class T>
inline int plan_deep_nt(GemmArgs<T>& a, int max_s) {
    long Mx = a.M, Nx = a.N, Kx = a.K;
    if (std::is_same<T, c64>::value) {
        if (!cplx_on_mfma<FORM_NT>(a.conjA, a.conjB, a.ext_ws)) return 1;
        Nx *= 2; Kx *= 2;
    } else if (!std::is_same<T, float>::value) {
        return 1;
    }
    if (Mx < CfgHuge::BM || Nx < CfgHuge::BN) return 1;
    const long wh = (long)ceil_div(Mx, CfgHuge::BM) * ceil_div(Nx, CfgHuge::BN);
    if (wh >= 192) return 1;
    const long S = (256 + wh - 1) / wh;
    if (S < 2 || S > max_s || Kx / S < 8192) return 1;
    const long kblocks = (Kx + 15) / 16;
    a.tile = TILE_HUGE;
    a.klen = (int)(((kblocks + S - 1) / S) * 16);
    a.ksplits = ceil_div(Kx, a.klen);
    a.split_planned = true;
    return a.ksplits;
}

// A/B knob of the pair schedule (TileCfg PIPE = 3): -1 = not yet read, 0 = pair schedule, 1 = plain schedule.
// Read ONCE from DCP_TN_PLAIN; dcp_debug_tn_plain() (test hook) overrides it so that one process can run both.
inline std::atomic<int>& tn_plain_flag() {
    static std::atomic<int> flag{-1};
    return flag;
}
inline bool tn_plain_schedule() {
    int v = tn_plain_flag().load(std::memory_order_relaxed);
    if (v < 0) {
        v = getenv("DCP_TN_PLAIN") != nullptr ? 1 : 0;
        tn_plain_flag().store(v, std::memory_order_relaxed);
    }
    return v != 0;
}

template <int FORM, class T, class Epi>
inline hipError_t gemm(hipStream_t stream, const GemmArgs<T>& a, const Epi& epi) {
    if constexpr (std::is_same<T, float>::value) {
        GemmProblem p;
        p.A = a.A; p.lda = a.lda; p.B = a.B; p.ldb = a.ldb;
        p.B2 = a.B2; p.ldb2 = a.ldb2; p.n_b1 = a.n_b1;
        p.A2 = a.A2; p.lda2 = a.lda2; p.m_a1 = a.m_a1;
        p.M = a.M; p.N = a.N; p.K = a.K;
        p.ksplits = a.ksplits; p.klen = a.klen;
        p.tiles_m = p.tiles_n = 0;
        // NT walks n tiles first (they share the A row panel); TN/NN walk m first.
        p.mt_fast = (FORM == FORM_NT) ? 0 : 1;
        constexpr int AL = (FORM == FORM_TN) ? XMAJOR : KMAJOR;
        constexpr int BL = (FORM == FORM_NT) ? KMAJOR : XMAJOR;
        int tier = pick_tier<FORM>(a.M, a.N, a.K, a.tile, a.split_planned, true);
        if (tier == TIER_SMALL && a.tile == TILE_AUTO) {
            // at most one workgroup per CU: occupancy is moot and the product is a chain of
            // load -> LDS -> MFMA round trips, one per K block; 64-deep blocks make 4x fewer of them
            // (small problems: Y.D^T 2048 x 32 x 512 31 -> 10 us).  Same summation order.
            const long wgs = (long)ceil_div(a.M, 64) * ceil_div(a.N, 64) * (a.ksplits > 1 ? a.ksplits : 1);
            const int depth = a.ksplits > 1 ? a.klen : a.K;
            if (wgs <= 256 && depth >= 256 && depth % 64 == 0 && a.K % 64 == 0) tier = TIER_SMALL_DEEP;
        }
        if (tier == TIER_SMALL) return launch_gemm_mfma<CfgSmall, AL, BL, Epi>(stream, p, epi);
        if (tier == TIER_SMALL_DEEP) return launch_gemm_mfma<CfgSmallDeep, AL, BL, Epi>(stream, p, epi);
        if constexpr (FORM == FORM_TN && (epi_mode<Epi>::value == 0)) {
            // reduction over samples, both panels row-contiguous: the pair schedule (DCP_TN_PLAIN=1: A/B knob)
            if (!tn_plain_schedule()) {
                if (tier == TIER_HUGE_DEEP) return launch_gemm_mfma<CfgHugeDeepX, AL, BL, Epi>(stream, p, epi);
                if (tier == TIER_LARGE) return launch_gemm_mfma<CfgLargeX, AL, BL, Epi>(stream, p, epi);
            }
        }
        if (tier == TIER_HUGE_DEEP) return launch_gemm_mfma<CfgHugeDeep, AL, BL, Epi>(stream, p, epi);
        if (tier == TIER_FLAT) return launch_gemm_mfma<CfgFlat, AL, BL, Epi>(stream, p, epi);
        if (tier == TIER_TALL) return launch_gemm_mfma<CfgTall, AL, BL, Epi>(stream, p, epi);
        if constexpr (FORM != FORM_TN) {
            if (tier == TIER_HUGE) return launch_gemm_mfma<CfgHuge, AL, BL, Epi>(stream, p, epi);
            if (tier == TIER_MID) return launch_gemm_mfma<CfgMid, AL, BL, Epi>(stream, p, epi);
        }
        return launch_gemm_mfma<CfgLarge, AL, BL, Epi>(stream, p, epi);
    } else {
        if constexpr (std::is_same<T, c64>::value) {
            if (cplx_on_mfma<FORM>(a.conjA, a.conjB, a.ext_ws)) {
                GemmProblem p;
                const float* Ar = reinterpret_cast<const float*>(a.A);
                p.A = Ar; p.lda = 2 * a.lda;
                p.B2 = nullptr; p.ldb2 = 0; p.n_b1 = 2 * a.N;
                p.ksplits = a.ksplits; p.klen = a.klen;
                p.tiles_m = p.tiles_n = 0;
                p.mt_fast = (FORM == FORM_NT) ? 0 : 1;
                constexpr int AL = (FORM == FORM_TN) ? XMAJOR : KMAJOR;
                constexpr int BL = (FORM == FORM_NT) ? KMAJOR : XMAJOR;
                if constexpr (FORM == FORM_TN) {
                    p.B = reinterpret_cast<const float*>(a.B); p.ldb = 2 * a.ldb;
                    if (a.B2 != nullptr) {
                        p.B2 = reinterpret_cast<const float*>(a.B2); p.ldb2 = 2 * a.ldb2;
                        p.n_b1 = 2 * a.n_b1;
                    }
                    p.M = 2 * a.M; p.N = 2 * a.N; p.K = a.K;
                    CplxTnEpi<Epi> ce{epi};
                    const int tier = pick_tier<FORM>(p.M, p.N, p.K, a.tile, a.split_planned);
                    if (tier == TIER_SMALL || tier == TIER_SMALL_DEEP || tier == TIER_FLAT || tier == TIER_TALL)
                        return launch_gemm_mfma<CfgSmall, AL, BL>(stream, p, ce);
                    if (!tn_plain_schedule()) return launch_gemm_mfma<CfgLargeX, AL, BL>(stream, p, ce);
                    return launch_gemm_mfma<CfgLarge, AL, BL>(stream, p, ce);
                } else if (cplx_planar_a<FORM>(a.M, a.N, a.conjA, a.conjB, a.ext_ws)) {
                    // rows(A): [2M, K] real image in the caller's scratch (2MK <= 4KN reals); B as it lies
                    if (a.A_rows != nullptr) {
                        p.A = a.A_rows; p.lda = a.lda_rows;
                    } else {
                        long g = ((long)a.M * a.K + 255) / 256;
                        if (g > 4096) g = 4096;
                        if (g < 1) g = 1;
                        hipLaunchKernelGGL((cplx_rows_kernel<float>), dim3((unsigned)g), dim3(256), 0, stream, a.A,
                                           (long)a.M, (long)a.K, a.lda, a.ext_ws);
                        p.A = a.ext_ws; p.lda = a.K;
                    }
                    p.B = reinterpret_cast<const float*>(a.B); p.ldb = 2 * a.ldb;
                    p.M = 2 * a.M; p.N = 2 * a.N; p.K = a.K;
                    if (a.ksplits <= 1) p.klen = 0;
                    CplxNnEpi<Epi> ce{epi};
                    const int tier = pick_tier<FORM>(p.M, p.N, p.K, a.tile, a.split_planned);
                    if (tier == TIER_SMALL_DEEP) return launch_gemm_mfma<CfgSmallDeep, AL, BL>(stream, p, ce);
                    if (tier == TIER_SMALL || tier == TIER_FLAT || tier == TIER_TALL)
                        return launch_gemm_mfma<CfgSmall, AL, BL>(stream, p, ce);
                    return launch_gemm_mfma<CfgLarge, AL, BL>(stream, p, ce);
                } else {
                    // ext(B): [2 rows(B), 2 cols(B)] real image in the caller's scratch
                    const long rowsB = (FORM == FORM_NT) ? a.N : a.K;
                    const long colsB = (FORM == FORM_NT) ? a.K : a.N;
                    long g = (rowsB * colsB + 255) / 256;
                    if (g > 4096) g = 4096;
                    if (g < 1) g = 1;
                    if (!a.ext_ready)
                        hipLaunchKernelGGL((cplx_ext_kernel<float>), dim3((unsigned)g), dim3(256), 0, stream, a.B,
                                           rowsB, colsB, a.ldb, a.ext_ws);
                    p.B = a.ext_ws; p.ldb = 2 * colsB;
                    p.M = a.M; p.N = 2 * a.N; p.K = 2 * a.K;
                    if (a.ksplits <= 1) p.klen = 0;
                    CplxColEpi<Epi> ce{epi};
                    const int tier = pick_tier<FORM>(p.M, p.N, p.K, a.tile, a.split_planned);
                    if (tier == TIER_SMALL || tier == TIER_SMALL_DEEP || tier == TIER_FLAT || tier == TIER_TALL)
                        return launch_gemm_mfma<CfgSmall, AL, BL>(stream, p, ce);
                    if (tier == TIER_HUGE) return launch_gemm_mfma<CfgHuge, AL, BL>(stream, p, ce);
                    if (tier == TIER_MID) return launch_gemm_mfma<CfgMid, AL, BL>(stream, p, ce);
                    return launch_gemm_mfma<CfgLarge, AL, BL>(stream, p, ce);
                }
            }
        }
        if constexpr (std::is_same<T, c128>::value) {
            const int Me = (FORM == FORM_TN) ? 2 * a.M : a.M;
            const int t128 = f64_tier(Me, 2 * a.N, a.tile);
            if (cplx_on_mfma<FORM>(a.conjA, a.conjB, a.ext_ws) && t128 != F64_GENERIC) {
                // complex128 on the fp64 MFMA core: same real-extended formulation as complex64
                GemmProblemD p;
                p.A = reinterpret_cast<const double*>(a.A); p.lda = 2 * a.lda;
                p.B2 = nullptr; p.ldb2 = 0; p.n_b1 = 2 * a.N;
                p.ksplits = a.ksplits; p.klen = a.klen;
                p.tiles_m = p.tiles_n = 0;
                p.mt_fast = (FORM == FORM_NT) ? 0 : 1;
                constexpr int AL = (FORM == FORM_TN) ? XMAJOR : KMAJOR;
                constexpr int BL = (FORM == FORM_NT) ? KMAJOR : XMAJOR;
                if constexpr (FORM == FORM_TN) {
                    p.B = reinterpret_cast<const double*>(a.B); p.ldb = 2 * a.ldb;
                    if (a.B2 != nullptr) {
                        p.B2 = reinterpret_cast<const double*>(a.B2); p.ldb2 = 2 * a.ldb2;
                        p.n_b1 = 2 * a.n_b1;
                    }
                    p.M = 2 * a.M; p.N = 2 * a.N; p.K = a.K;
                    CplxTnEpi<Epi, double> ce{epi};
                    return launch_f64_tier<AL, BL>(t128, stream, p, ce);
                } else if (cplx_planar_a<FORM>(a.M, a.N, a.conjA, a.conjB, a.ext_ws) &&
                           f64_tier(2 * a.M, 2 * a.N, a.tile) != F64_GENERIC) {
                    // planar rows of A against B's own memory, as for complex64 (mode 3)
                    if (a.A_rows != nullptr) {
                        p.A = a.A_rows; p.lda = a.lda_rows;
                    } else {
                        long g = ((long)a.M * a.K + 255) / 256;
                        if (g > 4096) g = 4096;
                        if (g < 1) g = 1;
                        hipLaunchKernelGGL((cplx_rows_kernel<double>), dim3((unsigned)g), dim3(256), 0, stream, a.A,
                                           (long)a.M, (long)a.K, a.lda, a.ext_ws);
                        p.A = a.ext_ws; p.lda = a.K;
                    }
                    p.B = reinterpret_cast<const double*>(a.B); p.ldb = 2 * a.ldb;
                    p.M = 2 * a.M; p.N = 2 * a.N; p.K = a.K;
                    if (a.ksplits <= 1) p.klen = 0;
                    CplxNnEpi<Epi, double> ce{epi};
                    return launch_f64_tier<AL, BL>(f64_tier(p.M, p.N, a.tile), stream, p, ce);
                } else {
                    const long rowsB = (FORM == FORM_NT) ? a.N : a.K;
                    const long colsB = (FORM == FORM_NT) ? a.K : a.N;
                    long g = (rowsB * colsB + 255) / 256;
                    if (g > 4096) g = 4096;
                    if (g < 1) g = 1;
                    if (!a.ext_ready)
                        hipLaunchKernelGGL((cplx_ext_kernel<double>), dim3((unsigned)g), dim3(256), 0, stream, a.B,
                                           rowsB, colsB, a.ldb, a.ext_ws);
                    p.B = a.ext_ws; p.ldb = 2 * colsB;
                    p.M = a.M; p.N = 2 * a.N; p.K = 2 * a.K;
                    if (a.ksplits <= 1) p.klen = 0;
                    CplxColEpi<Epi, double> ce{epi};
                    return launch_f64_tier<AL, BL>(t128, stream, p, ce);
                }
            }
        }
        if constexpr (std::is_same<T, double>::value) {
            const int t64 = f64_tier(a.M, a.N, a.tile, a.K);
            if (t64 != F64_GENERIC) {
                GemmProblemD p;
                p.A = a.A; p.lda = a.lda; p.B = a.B; p.ldb = a.ldb;
                p.B2 = a.B2; p.ldb2 = a.ldb2; p.n_b1 = a.n_b1;
                p.M = a.M; p.N = a.N; p.K = a.K;
                p.ksplits = a.ksplits; p.klen = a.klen;
                p.tiles_m = p.tiles_n = 0;
                p.mt_fast = (FORM == FORM_NT) ? 0 : 1;
                constexpr int AL = (FORM == FORM_TN) ? XMAJOR : KMAJOR;
                constexpr int BL = (FORM == FORM_NT) ? KMAJOR : XMAJOR;
                return launch_f64_tier<AL, BL>(t64, stream, p, epi);
            }
        }
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

inline bool al16_ptr(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <class T>
struct EpiStore {  // C[row, col] = acc
    // 16-byte epilogue available but OFF (interleaved in-process A/B, tools/vec_ab.py): x.D 16384x4096x256
    // 0.326 ms with it, 0.318 without: for a pure store the quad transposes cost as much VALU issue as the
    // narrower stores save.  It pays where the epilogue also LOADS per element (EpiMuNum: 1.045 -> 1.010 ms
    // on the dominant kernel).
    static constexpr bool kVec4 = false;
    T* C;
    long ldc;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        C[(long)r * ldc + c] = v;
    }
    bool vec_ok() const { return al16_ptr(C) && (ldc % 4) == 0; }
    __device__ __forceinline__ void vec4(int r, int c0, f32x4 v, int) const {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + (long)r * ldc + c0) = v;
    }
};

template <class T>
struct EpiSlab {  // split-K partial: slab[split][row, col] = acc
    static constexpr bool kVec4 = false;   // neutral in the A/B (1.134 vs 1.130 ms on x^T.Y): off
    T* slab;
    long ldc;
    long slab_stride;
    __device__ __forceinline__ void operator()(int r, int c, T v, int s) const {
        slab[(long)s * slab_stride + (long)r * ldc + c] = v;
    }
    bool vec_ok() const { return al16_ptr(slab) && (ldc % 4) == 0 && (slab_stride % 4) == 0; }
    __device__ __forceinline__ void vec4(int r, int c0, f32x4 v, int s) const {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(slab) + (long)s * slab_stride + (long)r * ldc + c0) = v;
    }
    // pair schedule of the fp32 MFMA core (TileCfg PIPE = 3): two adjacent columns in one 8-byte store
    static constexpr bool kVec2 = std::is_same<T, float>::value;
    bool vec2_ok() const {
        return (reinterpret_cast<uintptr_t>(slab) & 7) == 0 && (ldc % 2) == 0 && (slab_stride % 2) == 0;
    }
    __device__ __forceinline__ void store2(int r, int c0, float v0, float v1, int s) const {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f32x2*>(reinterpret_cast<float*>(slab) + (long)s * slab_stride + (long)r * ldc + c0) =
            f32x2{v0, v1};
    }
};

// out = cur * max(acc, 0) / max(den, 1e-15): acc is the POSITIVE gradient part
// (grads.py:84 with grad_pos produced by this GEMM).  den: full matrix (ld_den > 0)
// or one value per column (ld_den == 0), e.g. KL's colsum(D).
template <class T>
struct EpiMuNum {
    static constexpr bool kVec4 = std::is_same<T, float>::value;
    const T* cur;
    long ld_cur;
    const T* den;
    long ld_den;
    T* out;
    long ld_out;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        const T d = den[(long)r * ld_den + c];
        out[(long)r * ld_out + c] = cur[(long)r * ld_cur + c] * max_np(v, T(0)) /
                                    max_np(d, T(1.0e-15));
    }
    // (ld_den == 0: one denominator per column, read as 4 consecutive values of the vector)
    bool vec_ok() const {
        return al16_ptr(cur) && al16_ptr(den) && al16_ptr(out) && (ld_cur % 4) == 0 && (ld_den % 4) == 0 &&
               (ld_out % 4) == 0;
    }
    __device__ __forceinline__ void vec4(int r, int c0, f32x4 v, int) const {
        const f32x4 d = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(den) + (long)r * ld_den + c0);
        const f32x4 x = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(cur) + (long)r * ld_cur + c0);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = x[e] * max_np(v[e], 0.0f) / max_np(d[e], 1.0e-15f);
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (long)r * ld_out + c0) = o;
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
        out[(long)r * ld_out + c] = cur[(long)r * ld_cur + c] * max_np(nu, T(0)) /
                                    max_np(v, T(1.0e-15));
    }
};

// Split-K numerator: out = cur * max(sum_s slab_s, 0) / max(acc, 1e-15).  acc is the NEGATIVE part (x.G);
// the positive part (Y.D^T) arrives as S ordered split-K partials, summed here in the fixed order
// 0..S-1 (as mu_quotient_slabs_kernel does).  With few rows per GPU the Y.D^T product splits its
// reduction; the quotient then rides on the x.G product (16-byte epilogue) instead of a pass of its own.
template <class T>
struct EpiMuDenSlabs {
    static constexpr bool kVec4 = std::is_same<T, float>::value;
    const T* cur;
    long ld_cur;
    const T* slabs;      // [S][rows, ld_slab]
    long ld_slab;
    long slab_stride;
    int S;
    T* out;
    long ld_out;
    __device__ __forceinline__ void operator()(int r, int c, T v, int) const {
        const long o = (long)r * ld_slab + c;
        T nu = slabs[o];
        for (int s = 1; s < S; ++s) nu = nu + slabs[(long)s * slab_stride + o];
        out[(long)r * ld_out + c] = cur[(long)r * ld_cur + c] * max_np(nu, T(0)) / max_np(v, T(1.0e-15));
    }
    bool vec_ok() const {
        return al16_ptr(cur) && al16_ptr(slabs) && al16_ptr(out) && (ld_cur % 4) == 0 && (ld_slab % 4) == 0 &&
               (slab_stride % 4) == 0 && (ld_out % 4) == 0;
    }
    __device__ __forceinline__ void vec4(int r, int c0, f32x4 v, int) const {
        if constexpr (std::is_same<T, float>::value) {
            const long o = (long)r * ld_slab + c0;
            f32x4 nu = *reinterpret_cast<const f32x4*>(slabs + o);
            for (int s = 1; s < S; ++s) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(slabs + (long)s * slab_stride + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) nu[e] = nu[e] + p[e];
            }
            const f32x4 x = *reinterpret_cast<const f32x4*>(cur + (long)r * ld_cur + c0);
            f32x4 q;
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = x[e] * max_np(nu[e], 0.0f) / max_np(v[e], 1.0e-15f);
            *reinterpret_cast<f32x4*>(out + (long)r * ld_out + c0) = q;
        }
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
    static constexpr bool kVec4 = std::is_same<T, float>::value;   // one load per element: 16-byte form
    bool vec_ok() const { return al16_ptr(mask) && al16_ptr(out) && (ld_mask % 4) == 0 && (ld_out % 4) == 0; }
    __device__ __forceinline__ void vec4(int r, int c0, f32x4 v, int) const {
        if constexpr (std::is_same<T, float>::value) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(mask + (long)r * ld_mask + c0);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e] * m[e];
            *reinterpret_cast<f32x4*>(out + (long)r * ld_out + c0) = o;
        }
    }
};

// out = acc * mask for a BINARY mask handed over as row bits (see epi_rowbits in gemm_mfma_f32.hpp):
// the factor is exactly 1.0f or 0.0f and the product is formed as in EpiMulMask (so NaN / Inf / -0
// behave as the reference's  f * mask).  float only (the fp32 MFMA core implements the protocol).
struct EpiMulMaskBits {
    static constexpr bool kRowBits = true;
    const uint32_t* bits;   // [(rows + 31) / 32][ld_bits]
    long ld_bits;
    float* out;
    long ld_out;
    __device__ __forceinline__ uint32_t load_bits(int row0, int col) const {
        return bits[(long)(row0 >> 5) * ld_bits + col];
    }
    __device__ __forceinline__ void with_bit(int r, int c, float v, uint32_t bit, int) const {
        out[(long)r * ld_out + c] = v * (bit ? 1.0f : 0.0f);
    }
    __device__ __forceinline__ void operator()(int, int, float, int) const {}
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
    static constexpr bool kVec4 = std::is_same<T, float>::value;
    bool vec_ok() const {
        return al16_ptr(y) && al16_ptr(out) && (ld_y % 4) == 0 && (ld_out % 4) == 0 &&
               (mask == nullptr || (al16_ptr(mask) && (ld_mask % 4) == 0));
    }
    __device__ __forceinline__ void vec4(int r, int c0, f32x4 v, int) const {
        if constexpr (std::is_same<T, float>::value) {
            f32x4 yy = *reinterpret_cast<const f32x4*>(y + (long)r * ld_y + c0);
            if (mask != nullptr) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(mask + (long)r * ld_mask + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) yy[e] = yy[e] * m[e];
            }
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = yy[e] / (v[e] + 1.0e-15f);
            *reinterpret_cast<f32x4*>(out + (long)r * ld_out + c0) = o;
        }
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
