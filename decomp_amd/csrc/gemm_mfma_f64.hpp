// fp64 MFMA GEMM core for gfx950 (MI355X).  float64 is the reference's default dtype
// (NumPy arrays are float64 unless the caller says otherwise), so this is the path most
// drop-in users land on.
//
//   C(m, n) = sum_k A(m, k) * B(k, n)
//
// Instruction: v_mfma_f64_16x16x4_f64 -- 2048 flop, 64 cycles / SIMD on MI355X (fp64 matrix
// peak 78.6 TFLOP/s = the fp64 vector peak; what the matrix pipe buys is operand economy: one
// f64 of A and one of B per lane per 2048 flop, an 8x smaller LDS / register-file read rate than
// a 4x4 register-tiled v_fma_f64 loop).  Lane l: A[row l&15][k l>>4], B[k l>>4][col l&15];
// C/D: col = l&15, row = (l>>4) + 4*reg (NOT the f32 map, guide cdna_hip_programming.md).
//
// Workgroup: 128 x 128 tile, BK = 16, 1024 threads = 16 waves, each wave 32 x 32 = 2 x 2
// accumulators (32 VGPRs).  64 KiB LDS (two panels, double buffered).  Staging as in the fp32
// core: global -> registers -> LDS, one barrier per K block.
//
// LDS images:
//   KMAJOR panel: [rows][16] doubles, 16-byte chunks XOR-swizzled (chunk q of row r at
//     q ^ ((r/2) % 8)): the 16 rows of a ds_read_b128 lane group hit 16 distinct 4-bank slots.
//     Lane group g of MFMA step (c, j) consumes k = 8c + 2g + j, so one 16-byte read feeds two
//     steps (any fixed k permutation is a valid fma chain, same trick as the fp32 core).
//   XMAJOR panel: [16][rows + 16] doubles, read with ds_read_b64; the +16 pad puts the four lane
//     groups of a step on two disjoint bank halves (the 512-byte wave read needs two passes anyway).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_mfma_f32.hpp"  // KMAJOR / XMAJOR, xcd_remap, epi_mode

namespace dcp {

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

struct GemmProblemD {
    const double* A;
    long lda;
    const double* B;
    long ldb;
    const double* B2;
    long ldb2;
    int n_b1;
    int M, N, K;
    int ksplits, klen;
    int tiles_m, tiles_n;
    int tiles_n1;
    int mt_fast;
    int al_mask = 0;   // set by the launcher: bit 0 / 2 / 3 = A / B / B2 is 16-byte aligned with an even ld
};

template <int BM_, int BN_, int WM_, int WN_, int MINW_>
struct F64Cfg {
    static constexpr int BM = BM_, BN = BN_, BK = 16, WM = WM_, WN = WN_, MINW = MINW_;
    static constexpr int NWAVES = (BM / WM) * (BN / WN);
    static constexpr int NT = 64 * NWAVES;
    static constexpr int XPAD = 16;
};
// The production tile: 16 waves of 32 x 32 (measured on Y.D^T, 32768x256x4096: 64.4 TF = 82 % of
// the fp64 peak; 8 waves of 32 x 64: 55.9; 4 waves of 64 x 64: 59.2; 128 x 256 / 16 waves: 64.7 --
// tools/bench_f64.py).  One workgroup already puts 4 waves on every SIMD, so thin problems with
// one tile per CU keep the matrix pipe fed.
typedef F64Cfg<128, 128, 32, 32, 1> F64Tile;

template <int LAY, int ROWS, int NT_ = F64Tile::NT>
struct PanelD {
    static constexpr int BK = 16;
    static constexpr int XSTRIDE = ROWS + 16;
    static constexpr int ELEMS = (LAY == KMAJOR) ? ROWS * BK : BK * XSTRIDE;
    static constexpr int CH = ROWS * BK / 2 / NT_;   // 16-byte chunks per thread per block (checked by the loaders)
    static constexpr bool WHOLE = (ROWS * BK % (2 * NT_) == 0) && CH > 0;
    // double offset of 16-byte chunk q (2 doubles) of row r in the swizzled KMAJOR image
    __device__ static __forceinline__ int kchunk(int r, int q) {
        return r * BK + ((q ^ ((r >> 1) & 7)) << 1);
    }
};

// EDGE as in the fp32 core (panel_gload): `vec` (wave-uniform) = operand 16-byte aligned, whole K block
// inside, row-contiguous panels only with the whole tile inside -> 16-byte loads with the row clamped;
// otherwise element-wise loads from clamped addresses and a select (no load under a lane predicate).
template <int LAY, int ROWS, bool EDGE, int NT_, int CH>
__device__ __forceinline__ void panel_gload_d(f64x2 (&r)[CH], const double* __restrict__ p, long ld,
                                              int row0, int nrows, int k0, int kend, int tid,
                                              bool vec = false) {
    static_assert(CH == PanelD<LAY, ROWS, NT_>::CH, "register panel size");
    static_assert(PanelD<LAY, ROWS, NT_>::WHOLE, "panel must be whole chunks per thread");
    constexpr int BK = 16;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int idx = tid + i * NT_;
        if (LAY == KMAJOR) {
            const int row = idx / (BK / 2), kq = (idx % (BK / 2)) * 2;
            if (!EDGE) {
                r[i] = *reinterpret_cast<const f64x2*>(p + (long)(row0 + row) * ld + (k0 + kq));
            } else {
                const bool rok = (row0 + row) < nrows;
                const double* prow = p + (long)(rok ? row0 + row : nrows - 1) * ld;
                if (vec) {
                    r[i] = *reinterpret_cast<const f64x2*>(prow + (k0 + kq));
                } else {
                    double v[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) v[e] = prow[min(k0 + kq + e, kend - 1)];
#pragma unroll
                    for (int e = 0; e < 2; ++e) r[i][e] = (rok && (k0 + kq + e) < kend) ? v[e] : 0.0;
                }
            }
        } else {
            const int kr = idx / (ROWS / 2), rq = (idx % (ROWS / 2)) * 2;
            if (!EDGE) {
                r[i] = *reinterpret_cast<const f64x2*>(p + (long)(k0 + kr) * ld + (row0 + rq));
            } else {
                const bool kok = (k0 + kr) < kend;
                const double* pk = p + (long)(kok ? k0 + kr : kend - 1) * ld;
                if (vec) {
                    r[i] = *reinterpret_cast<const f64x2*>(pk + (row0 + rq));
                } else {
                    double v[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) v[e] = pk[min(row0 + rq + e, nrows - 1)];
#pragma unroll
                    for (int e = 0; e < 2; ++e) r[i][e] = (kok && (row0 + rq + e) < nrows) ? v[e] : 0.0;
                }
            }
        }
    }
}

template <int LAY, int ROWS, int NT_, int CH>
__device__ __forceinline__ void panel_lds_store_d(double* s, const f64x2 (&r)[CH], int tid) {
    static_assert(CH == PanelD<LAY, ROWS, NT_>::CH, "register panel size");
    constexpr int BK = 16;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int idx = tid + i * NT_;
        if (LAY == KMAJOR) {
            const int row = idx / (BK / 2), q = idx % (BK / 2);
            *reinterpret_cast<f64x2*>(s + PanelD<LAY, ROWS>::kchunk(row, q)) = r[i];
        } else {
            const int kr = idx / (ROWS / 2), rq = (idx % (ROWS / 2)) * 2;
            *reinterpret_cast<f64x2*>(s + kr * PanelD<LAY, ROWS>::XSTRIDE + rq) = r[i];
        }
    }
}

// Operands of MFMA steps (c, 0) and (c, 1) for this lane: element (row, k = 8c + 2g + j) in [j].
template <int LAY, int ROWS>
__device__ __forceinline__ f64x2 panel_frag_d(const double* s, int row, int c, int g) {
    if (LAY == KMAJOR) {
        return *reinterpret_cast<const f64x2*>(s + PanelD<LAY, ROWS>::kchunk(row, 4 * c + g));
    } else {
        f64x2 v;
        const double* q = s + (8 * c + 2 * g) * PanelD<LAY, ROWS>::XSTRIDE + row;
        v[0] = q[0];
        v[1] = q[PanelD<LAY, ROWS>::XSTRIDE];
        return v;
    }
}

template <class Cfg, int ALAY, int BLAY, bool EDGE, class Epi>
__global__ void __launch_bounds__(Cfg::NT, Cfg::MINW) gemm_mfma_f64_kernel(GemmProblemD p, Epi epi) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    constexpr int WM = Cfg::WM, WN = Cfg::WN;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int WAVES_N = BN / WN;
    constexpr int NT = Cfg::NT;
    typedef PanelD<ALAY, BM, NT> GA;
    typedef PanelD<BLAY, BN, NT> GB;
    // panels above the 64 KiB static limit (XMAJOR pads) use dynamic LDS, raised by the launcher
    constexpr int LDS_DOUBLES = 2 * (GA::ELEMS + GB::ELEMS);
    constexpr bool DYN = (LDS_DOUBLES * 8 > 65536);
    __shared__ __attribute__((aligned(16))) double smem_static[DYN ? 2 : LDS_DOUBLES];
    extern __shared__ __attribute__((aligned(16))) double smem_dyn_d[];
    double* smem = DYN ? smem_dyn_d : smem_static;
    double* sA0 = smem;
    double* sB0 = smem + 2 * GA::ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l15 = lane & 15, g = lane >> 4;

    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.tiles_m * p.tiles_n;
    const int split = lid / tiles;
    const int t = lid - split * tiles;
    int mt, nt;
    if (p.mt_fast) {
        mt = t % p.tiles_m;
        nt = t / p.tiles_m;
    } else {
        nt = t % p.tiles_n;
        mt = t / p.tiles_n;
    }
    const int m0 = mt * BM;
    const int kbeg = split * p.klen;
    const int kend = min(p.K, kbeg + p.klen);

    const double* Bp = p.B;
    long ldb = p.ldb;
    int nB0 = nt * BN;
    int nBrows = p.n_b1;
    int n0 = nB0;
    int ncol_end = p.n_b1;
    if (nt >= p.tiles_n1) {
        Bp = p.B2;
        ldb = p.ldb2;
        nB0 = (nt - p.tiles_n1) * BN;
        nBrows = p.N - p.n_b1;
        n0 = p.n_b1 + nB0;
        ncol_end = p.N;
    }

    // bounds-checked instantiation: may this tile's panels still use 16-byte loads? (see panel_gload_d)
    const bool a_al = EDGE && (p.al_mask & 1) && (ALAY == KMAJOR || m0 + BM <= p.M);
    const bool b_al = EDGE && ((p.al_mask >> (nt >= p.tiles_n1 ? 3 : 2)) & 1) &&
                      (BLAY == KMAJOR || nB0 + BN <= nBrows);

    f64x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

    f64x2 ra[GA::CH], rb[GB::CH];
    const int nkb = (kend - kbeg + BK - 1) / BK;
    if (nkb > 0) {
        panel_gload_d<ALAY, BM, EDGE, NT>(ra, p.A, p.lda, m0, p.M, kbeg, kend, tid, a_al && (kbeg) + BK <= kend);
        panel_gload_d<BLAY, BN, EDGE, NT>(rb, Bp, ldb, nB0, nBrows, kbeg, kend, tid, b_al && (kbeg) + BK <= kend);
        panel_lds_store_d<ALAY, BM, NT>(sA0, ra, tid);
        panel_lds_store_d<BLAY, BN, NT>(sB0, rb, tid);
    }
    __syncthreads();

    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        const double* sA = sA0 + cur * GA::ELEMS;
        const double* sB = sB0 + cur * GB::ELEMS;
        const bool more = (kb + 1) < nkb;
        if (more) {
            const int k0 = kbeg + (kb + 1) * BK;
            panel_gload_d<ALAY, BM, EDGE, NT>(ra, p.A, p.lda, m0, p.M, k0, kend, tid, a_al && (k0) + BK <= kend);
            panel_gload_d<BLAY, BN, EDGE, NT>(rb, Bp, ldb, nB0, nBrows, k0, kend, tid, b_al && (k0) + BK <= kend);
        }
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            f64x2 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = panel_frag_d<ALAY, BM>(sA, wm * WM + i * 16 + l15, c, g);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = panel_frag_d<BLAY, BN>(sB, wn * WN + j * 16 + l15, c, g);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i][s], fb[j][s], acc[i][j],
                                                                         0, 0, 0);
        }
        if (more) {
            panel_lds_store_d<ALAY, BM, NT>(sA0 + (cur ^ 1) * GA::ELEMS, ra, tid);
            panel_lds_store_d<BLAY, BN, NT>(sB0 + (cur ^ 1) * GB::ELEMS, rb, tid);
        }
        __syncthreads();
    }

    // C layout of the 16x16 f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg.
    // Epi::kMode as in the fp32 core: 0 real; 1 complex product on real-extended operands, the
    // (re, im) of an output sit in lanes 2q / 2q+1; 2 complex x^H y on the real views of both
    // operands, the 2x2 block (rr ri; ir ii) sits in lanes (g, 2q), (g, 2q+1), (g+1, 2q),
    // (g+1, 2q+1) of one register: re = rr + ii, im = ri - ir.
    constexpr int MODE = epi_mode<Epi>::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * WM + i * 16 + g + 4 * r;
                const double v = acc[i][j][r];
                if constexpr (MODE == 0) {
                    if (!EDGE || (row < p.M && col < ncol_end)) epi(row, col, v, split);
                } else if constexpr (MODE == 1) {
                    const double o = __shfl_xor(v, 1, 64);          // executed by every lane
                    if (!(lane & 1) && (!EDGE || (row < p.M && col < ncol_end)))
                        epi.pair(row, col >> 1, v, o, split);
                } else {
                    // mode 2 (x^H y): re = rr + ii, im = ri - ir ;  mode 3 (planar rows of x times y): re = rr - ii, im = ri + ir
                    const double ii = __shfl_xor(v, 17, 64);
                    const double ri = __shfl_xor(v, 1, 64);
                    const double ir = __shfl_xor(v, 16, 64);
                    if (!(lane & 1) && !(g & 1) && (!EDGE || (row < p.M && col < ncol_end))) {
                        if constexpr (MODE == 2) epi.pair(row >> 1, col >> 1, v + ii, ri - ir, split);
                        else epi.pair(row >> 1, col >> 1, v - ii, ri + ir, split);
                    }
                }
            }
        }
    }
}

template <class Cfg, int ALAY, int BLAY, class Epi>
inline hipError_t launch_gemm_mfma_f64_cfg(hipStream_t stream, GemmProblemD p, const Epi& epi) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    if (p.B2 == nullptr) p.n_b1 = p.N;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n1 = (p.n_b1 + BN - 1) / BN;
    p.tiles_n = p.tiles_n1 + (p.N - p.n_b1 + BN - 1) / BN;
    if (p.ksplits < 1) p.ksplits = 1;
    if (p.ksplits == 1) {
        p.klen = ((p.K + BK - 1) / BK) * BK;
        if (p.klen == 0) p.klen = BK;
    }
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    bool fast = (p.M % BM == 0) && (p.N % BN == 0) && (p.K % BK == 0) && (p.klen % BK == 0) &&
                (p.lda % 2 == 0) && (p.ldb % 2 == 0) && al16(p.A) && al16(p.B) && p.K > 0;
    if (p.B2 != nullptr) fast = fast && (p.n_b1 % BN == 0) && (p.ldb2 % 2 == 0) && al16(p.B2);
    p.al_mask = ((al16(p.A) && p.lda % 2 == 0) ? 1 : 0) | ((al16(p.B) && p.ldb % 2 == 0) ? 4 : 0) |
                ((p.B2 != nullptr && al16(p.B2) && p.ldb2 % 2 == 0) ? 8 : 0);
    const int grid = p.tiles_m * p.tiles_n * p.ksplits;
    if (grid <= 0) return hipSuccess;
    constexpr int lds_bytes = 16 * (PanelD<ALAY, BM, Cfg::NT>::ELEMS + PanelD<BLAY, BN, Cfg::NT>::ELEMS);
    constexpr int dyn_bytes = lds_bytes > 65536 ? lds_bytes : 0;
    if (dyn_bytes) {
        static DynLdsRaised raised_fast, raised_edge;   // per instantiation
        std::atomic<bool>& raised = fast ? raised_fast.on_current_device() : raised_edge.on_current_device();
        if (!raised) {
            const void* fn = fast ? reinterpret_cast<const void*>(&gemm_mfma_f64_kernel<Cfg, ALAY, BLAY, false, Epi>)
                                  : reinterpret_cast<const void*>(&gemm_mfma_f64_kernel<Cfg, ALAY, BLAY, true, Epi>);
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, dyn_bytes);
            if (e != hipSuccess) return e;
            raised = true;
        }
    }
    if (fast)
        hipLaunchKernelGGL((gemm_mfma_f64_kernel<Cfg, ALAY, BLAY, false, Epi>), dim3(grid), dim3(Cfg::NT),
                           dyn_bytes, stream, p, epi);
    else
        hipLaunchKernelGGL((gemm_mfma_f64_kernel<Cfg, ALAY, BLAY, true, Epi>), dim3(grid), dim3(Cfg::NT),
                           dyn_bytes, stream, p, epi);
    return hipGetLastError();
}

template <int ALAY, int BLAY, class Epi>
inline hipError_t launch_gemm_mfma_f64(hipStream_t stream, GemmProblemD p, const Epi& epi) {
    return launch_gemm_mfma_f64_cfg<F64Tile, ALAY, BLAY, Epi>(stream, p, epi);
}

}  // namespace dcp
