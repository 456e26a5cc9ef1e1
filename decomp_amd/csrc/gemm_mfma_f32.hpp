// fp32 MFMA GEMM core for gfx950 (MI355X), written for 64-wide wavefronts.
//
//   C(m, n) = sum_k A(m, k) * B(k, n)        (reduction index k)
//
// Instruction: v_mfma_f32_32x32x2_f32 (exact f32, 64 cycles / SIMD / instr).
// A workgroup is 256 threads = 4 waves; each wave owns a WM x WN sub-tile made
// of (WM/32) x (WN/32) MFMA 32x32 accumulators.  Operand panels are staged
// global -> registers -> LDS (double buffered, one barrier per K block) so the
// next block's HBM/L2 latency hides under the current block's MFMAs.
//
// Operand memory layouts (both for A and for B):
//   KMAJOR : element (row, k) at p[row * ld + k]   (reduction index contiguous)
//   XMAJOR : element (row, k) at p[k * ld + row]   (row index contiguous)
// "row" is m for A and n for B.  So
//   NT  (A KMAJOR, B KMAJOR):  Y . D^T,  D . D^T,  y . A^H
//   NN  (A KMAJOR, B XMAJOR):  x . G,  x . D,  S . D
//   TN  (A XMAJOR, B XMAJOR):  x^T . Y,  x^T . x   (reduction over samples)
//
// LDS images:
//   KMAJOR panel: [rows][BK] floats, no padding, 16-byte chunks XOR-swizzled:
//     chunk q of row r lives at chunk q ^ ((r / R) % C), C = BK/4 chunks per row,
//     R = 64/BK rows per 256-byte bank row.  Each lane fetches 4 consecutive k of
//     its row with one ds_read_b128; the 16 rows of a b128 lane group then hit 16
//     distinct 4-bank slots (conflict free), and a 128x128 tile pair needs 32 KiB
//     so four workgroups fit a CU's 160 KiB.
//   XMAJOR panel: [BK][rows] floats, read with ds_read_b32 (32 consecutive
//     dwords per half wave: conflict free).
// k order inside a K block: the lane half h = lane >> 5 of MFMA step (c, j)
// consumes k = 8c + 4h + j.  Any fixed permutation of k is a valid fp32 fma
// chain; this one lets a KMAJOR lane read its 4 k's as one 16-byte load.
//
// Split-K: grid = tiles_m * tiles_n * ksplits; split s reduces
// k in [s*klen, min(K, (s+1)*klen)).  The epilogue receives the split index.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>

namespace dcp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { KMAJOR = 0, XMAJOR = 1 };

struct GemmProblem {
    const float* A;
    long lda;
    const float* B;   // columns [0, n_b1)
    long ldb;
    const float* B2;  // columns [n_b1, N) (optional second segment), may be null
    long ldb2;
    int n_b1;
    const float* A2 = nullptr;  // rows [m_a1, M) of A (optional second ROW segment: two products against
    long lda2 = 0;              // one B in a single launch, e.g. [f ; Y o M] . D^T), may be null
    int m_a1 = 0;
    int M, N, K;
    int ksplits, klen;
    int tiles_m, tiles_n;
    int tiles_n1;  // n tiles that belong to the first B segment (tiles never straddle)
    int tiles_m1;  // m tiles that belong to the first A segment
    int mt_fast;   // 1: consecutive logical ids walk m tiles first
    int vec_epi;   // set by the launcher: the functor's 16-byte epilogue (vec4) may be used
    int al_mask;   // set by the launcher: bit 0 / 1 / 2 / 3 = A / A2 / B / B2 is 16-byte aligned with ld % 4 == 0
};

// PIPE_ = 1: the per-block barrier sits between the MFMA groups of a K block (operands of the
// last group already in registers, first group of the next block prefetched right after the
// barrier), so no wave leaves the barrier with nothing to feed the matrix pipe.
// PIPE_ = 3 (both panels row-contiguous, i.e. the reduction-over-samples form, 64 x 64 wave tiles): the
// "pair" schedule.  A lane reads TWO adjacent rows of a panel line with one ds_read_b64 (256 B/clk, against
// 128 B/clk for the ds_read2_b32 pairs the plain schedule compiles to) and owns the interleaved rows
// 2 l + i / columns 2 l + j of the wave tile instead of l + 32 i / l + 32 j; MFMA step s takes k = 2 s + h.
// Fragment reads run TWO steps ahead of the MFMAs that consume them (the plain schedule reads a step's
// operands and waits for them right before its four MFMAs: one exposed LDS round trip per step).  The
// epilogue stores the two adjacent columns of a lane with one 8-byte store; complex products (mode 2) find
// the whole 2 x 2 real block of a complex output in ONE lane: no shuffles, every lane stores.
// MF_ = 32: v_mfma_f32_32x32x2_f32; MF_ = 16: v_mfma_f32_16x16x4_f32 (same flop rate, same LDS
// read volume and accumulator count; the chip may hold a different clock on it, guide rule 28).
template <int BM_, int BN_, int BK_, int WM_, int WN_, int MINW_, int PIPE_ = 0, int MF_ = 32>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, MINW = MINW_;
    static constexpr int PIPE = PIPE_;
    static constexpr int MF = MF_;
    static constexpr int NWAVES = (BM_ / WM_) * (BN_ / WN_);
    static constexpr int NTHREADS = 64 * NWAVES;
};

// blocks b and b+8 share an XCD (round-robin dispatch).  Give every XCD a
// contiguous run of logical ids so neighbouring tiles share that XCD's L2.
// Bijective for any n (guide T1).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, within = id >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + within;
}

// Epilogue functors may carry `static constexpr int kMode` (see the kernel's epilogue).
template <class E, class = void>
struct epi_mode { static constexpr int value = 0; };
template <class E>
struct epi_mode<E, decltype((void)E::kMode)> { static constexpr int value = E::kMode; };

// Epilogue functors may carry `static constexpr bool kRowBits = true`: a per-element 0/1 factor
// packed 32 ROWS to a word (word (row / 32, col) holds rows 32*(row/32) .. +31 of column col).
// The kernel then fetches, BEFORE its main loop, the one word per 32x32 accumulator a lane needs
// (its column, the accumulator's 32 rows: TM x TN registers) and hands every element its bit:
//   uint32_t load_bits(int row0 /* multiple of 32 */, int col)       epi.with_bit(row, col, v, bit, split)
// so that the epilogue issues no loads of the factor at all (a float mask costs one 4-byte load per
// element there, in batches of dependent round trips: registers are full).
template <class E, class = void>
struct epi_rowbits { static constexpr bool value = false; };
template <class E>
struct epi_rowbits<E, decltype((void)E::kRowBits)> { static constexpr bool value = E::kRowBits; };

// Epilogue functors may carry `static constexpr bool kVec4 = true` together with
//     bool vec_ok() const                      (host: every array 16-byte aligned, leading dims % 4 == 0)
//     void vec4(int row, int col0, f32x4 v, int split) const      (columns col0 .. col0 + 3 of one row)
// The 32x32 accumulator holds a COLUMN per lane (4 consecutive rows in registers 4g .. 4g+3); a 4x4
// transpose inside every lane quad (two DPP quad_perm exchanges) turns that into 4 consecutive columns
// of one row per lane, so the epilogue moves 16 bytes per lane and instruction: 4x fewer loads / stores
// and address computations than the per-element form (the epilogue is store-issue bound, guide T21).
template <class E, class = void>
struct epi_vec4 { static constexpr bool value = false; };
template <class E>
struct epi_vec4<E, decltype((void)E::kVec4)> { static constexpr bool value = E::kVec4; };

// Pair-schedule functors may carry `static constexpr bool kVec2 = true` with
//     bool vec2_ok() const                                   (host: base 8-byte aligned, leading dims even)
//     void store2(int row, int col0, float v0, float v1, int split) const     (columns col0, col0 + 1; col0 even)
template <class E, class = void>
struct epi_vec2 { static constexpr bool value = false; };
template <class E>
struct epi_vec2<E, decltype((void)E::kVec2)> { static constexpr bool value = E::kVec2; };
template <class E>
inline bool epi_vec2_ok(const E& e) {
    if constexpr (epi_vec2<E>::value) return e.vec2_ok();
    else return false;
}

template <class E>
inline bool epi_vec_ok(const E& e) {
    if constexpr (epi_vec4<E>::value) return e.vec_ok();
    else return false;
}

template <int CTRL>
__device__ __forceinline__ float quad_perm_f32(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
// x[j] = element (row j, column of lane t) of a 4x4 block held by the lane quad  ->  element (row t, column j)
__device__ __forceinline__ f32x4 quad_transpose(f32x4 x, int t) {
    const bool odd = t & 1, hi = t & 2;
    const float n0 = quad_perm_f32<0xB1>(x[0]), n1 = quad_perm_f32<0xB1>(x[1]);      // lanes t ^ 1
    const float n2 = quad_perm_f32<0xB1>(x[2]), n3 = quad_perm_f32<0xB1>(x[3]);
    f32x4 z;
    z[0] = odd ? n1 : x[0];
    z[1] = odd ? x[1] : n0;
    z[2] = odd ? n3 : x[2];
    z[3] = odd ? x[3] : n2;
    const float m0 = quad_perm_f32<0x4E>(z[0]), m1 = quad_perm_f32<0x4E>(z[1]);      // lanes t ^ 2
    const float m2 = quad_perm_f32<0x4E>(z[2]), m3 = quad_perm_f32<0x4E>(z[3]);
    f32x4 y;
    y[0] = hi ? m2 : z[0];
    y[1] = hi ? m3 : z[1];
    y[2] = hi ? z[2] : m0;
    y[3] = hi ? z[3] : m1;
    return y;
}

template <int LAY, int ROWS, int BK, int NT = 256, int MF = 32>
struct PanelGeom {
    static constexpr int STRIDE = (LAY == KMAJOR) ? BK : ROWS;
    static constexpr int CHUNKS = BK / 4;   // 16-byte chunks per KMAJOR row
    static constexpr int RPB = 64 / BK;     // KMAJOR rows per 256-byte bank row
    static_assert(BK <= 64 && 64 % BK == 0, "BK must divide 64");
    static_assert(MF == 32 || BK == 16, "the 16x16x4 lane map is laid out for BK = 16");
    // float offset of chunk q (4 floats) of row r in the swizzled KMAJOR image
    __device__ static __forceinline__ int kchunk(int r, int q) {
        if (MF == 16) {
            // 16x16x4 map: lane = 16 * q + row (k quarter q of a 16-row fragment).  The four rows
            // r, r+4, r+8, r+12 share a 64-byte column of the 256-byte bank row; XOR with
            // f(r/4 % 4), f = (0, 2, 3, 1), sends the 16 lanes of every ds_read_b128 group to 16
            // distinct 16-byte slots (checked against the group lists of MI355X_MICROARCH.md).
            const int f = (0x78 >> (2 * ((r >> 2) & 3))) & 3;      // 0b01'11'10'00 -> f = 0, 2, 3, 1
            return r * BK + ((q ^ f) << 2);
        }
        return r * BK + ((q ^ ((r / RPB) % CHUNKS)) << 2);
    }
    static constexpr int LINES = (LAY == KMAJOR) ? ROWS : BK;
    static constexpr int ELEMS = STRIDE * LINES;
    static constexpr int F4 = ROWS * BK / 4 / NT;  // float4 per thread per block
    static_assert(ROWS * BK % (4 * NT) == 0, "panel must be a whole number of float4 per thread");
};

// Loads one panel block into registers.
// EDGE (ragged / unaligned problems): `vec` (wave-uniform) says that this operand is 16-byte aligned, the
// whole K block lies inside the reduction range and -- for row-contiguous panels -- the whole tile lies
// inside the matrix: then the load is the fast path's 16-byte load with the ROW index clamped (rows past
// the end fetch a duplicate; they only feed accumulator rows / columns the epilogue never stores).
// Otherwise every element is fetched on its own from a clamped address and zeroed by a select: no load
// sits under a lane predicate (a predicated load compiles to a branch and a full wait of its own).
template <int LAY, int ROWS, int BK, bool EDGE, int NT, int F4>
__device__ __forceinline__ void panel_gload(f32x4 (&r)[F4],
                                            const float* __restrict__ p, long ld, int row0,
                                            int nrows, int k0, int kend, int tid, bool vec = false) {
    static_assert(F4 == PanelGeom<LAY, ROWS, BK, NT>::F4, "register panel size");
#pragma unroll
    for (int i = 0; i < F4; ++i) {
        const int idx = tid + i * NT;
        if (LAY == KMAJOR) {
            const int row = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
            if (!EDGE) {
                r[i] = *reinterpret_cast<const f32x4*>(p + (long)(row0 + row) * ld + (k0 + kq));
            } else {
                const bool rok = (row0 + row) < nrows;
                const float* prow = p + (long)(rok ? row0 + row : nrows - 1) * ld;
                if (vec) {
                    r[i] = *reinterpret_cast<const f32x4*>(prow + (k0 + kq));
                } else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = prow[min(k0 + kq + e, kend - 1)];
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[i][e] = (rok && (k0 + kq + e) < kend) ? v[e] : 0.0f;
                }
            }
        } else {
            const int kr = idx / (ROWS / 4), rq = (idx % (ROWS / 4)) * 4;
            if (!EDGE) {
                r[i] = *reinterpret_cast<const f32x4*>(p + (long)(k0 + kr) * ld + (row0 + rq));
            } else {
                const bool kok = (k0 + kr) < kend;
                const float* pk = p + (long)(kok ? k0 + kr : kend - 1) * ld;
                if (vec) {
                    r[i] = *reinterpret_cast<const f32x4*>(pk + (row0 + rq));
                } else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = pk[min(row0 + rq + e, nrows - 1)];
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[i][e] = (kok && (row0 + rq + e) < nrows) ? v[e] : 0.0f;
                }
            }
        }
    }
}

template <int LAY, int ROWS, int BK, int NT, int MF = 32, int F4>
__device__ __forceinline__ void panel_lds_store(float* s, const f32x4 (&r)[F4], int tid) {
    static_assert(F4 == PanelGeom<LAY, ROWS, BK, NT>::F4, "register panel size");
    constexpr int STRIDE = PanelGeom<LAY, ROWS, BK>::STRIDE;
#pragma unroll
    for (int i = 0; i < F4; ++i) {
        const int idx = tid + i * NT;
        if (LAY == KMAJOR) {
            const int row = idx / (BK / 4), q = idx % (BK / 4);
            *reinterpret_cast<f32x4*>(s + PanelGeom<LAY, ROWS, BK, 256, MF>::kchunk(row, q)) = r[i];
        } else {
            const int kr = idx / (ROWS / 4), rq = (idx % (ROWS / 4)) * 4;
            *reinterpret_cast<f32x4*>(s + kr * STRIDE + rq) = r[i];
        }
    }
}

// Fragment for MFMA steps (c, 0..3): lane (l31, h) gets element (row, k=8c+4h+j) in [j].
template <int LAY, int ROWS, int BK>
__device__ __forceinline__ f32x4 panel_frag(const float* s, int row, int c, int h) {
    constexpr int STRIDE = PanelGeom<LAY, ROWS, BK>::STRIDE;
    if (LAY == KMAJOR) {
        return *reinterpret_cast<const f32x4*>(s + PanelGeom<LAY, ROWS, BK>::kchunk(row, 2 * c + h));
    } else {
        f32x4 v;
        const float* q = s + (c * 8 + 4 * h) * STRIDE + row;
        v[0] = q[0];
        v[1] = q[STRIDE];
        v[2] = q[2 * STRIDE];
        v[3] = q[3 * STRIDE];
        return v;
    }
}

// 16x16x4 form: fragment of a 16-row group for the 4 MFMA steps of one 16-deep K block.  Lane
// (l15 = lane & 15, q = lane >> 4) gets element (row, k = 4q + j) in [j]: step j consumes
// k = {j, 4 + j, 8 + j, 12 + j} on the four lane quarters.
template <int LAY, int ROWS, int BK>
__device__ __forceinline__ f32x4 panel_frag16(const float* s, int row, int q) {
    constexpr int STRIDE = PanelGeom<LAY, ROWS, BK>::STRIDE;
    if (LAY == KMAJOR) {
        return *reinterpret_cast<const f32x4*>(s + PanelGeom<LAY, ROWS, BK, 256, 16>::kchunk(row, q));
    } else {
        f32x4 v;
        const float* t = s + (4 * q) * STRIDE + row;
        v[0] = t[0];
        v[1] = t[STRIDE];
        v[2] = t[2 * STRIDE];
        v[3] = t[3 * STRIDE];
        return v;
    }
}

// ---- pair schedule (TileCfg PIPE = 3): hand-placed LDS reads ---------------------------------------------
// hipcc merges adjacent 8-byte LDS reads into ds_read2st64_b64 (32-bank mode: lanes l and l + 16 collide) and
// parks every read directly in front of its consumers behind s_waitcnt lgkmcnt(0), whatever the source order
// says; so the fragment reads are inline asm (plain ds_read_b64, 64-bank mode, conflict free for 64
// consecutive dwords per half wave) and the counted waits are ours (guide 5.7 item 1, form ii: the wait
// statement names the destinations "+v", which keeps their consumers behind it).
typedef float pair_f32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ void lds_read_b64_at(pair_f32x2& d, unsigned lds_addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(lds_addr), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_pair(pair_f32x2& a, pair_f32x2& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const float* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const float*)p;
}
// steps ST .. NS-1 of one K block: operands of step ST + 2 are requested before the MFMAs of step ST issue
template <int ST, int NS, int ASTEP_BYTES, int BSTEP_BYTES>
struct PairSteps {
    __device__ static __forceinline__ void run(f32x16 (&acc)[2][2], pair_f32x2& a0, pair_f32x2& b0, pair_f32x2& a1,
                                               pair_f32x2& b1, unsigned aaddr, unsigned baddr) {
        pair_f32x2 a2 = a1, b2 = b1;
        if constexpr (ST + 2 < NS) {
            lds_read_b64_at<(ST + 2) * ASTEP_BYTES>(a2, aaddr);
            lds_read_b64_at<(ST + 2) * BSTEP_BYTES>(b2, baddr);
        }
        // reads issued after those of step ST: two steps' worth, less at the end of the block
        constexpr int NEWER = (ST + 2 < NS) ? 4 : ((ST + 1 < NS) ? 2 : 0);
        lds_wait_pair<NEWER>(a0, b0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0[0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0[1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0[0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0[1], acc[1][1], 0, 0, 0);
        if constexpr (ST + 1 < NS) PairSteps<ST + 1, NS, ASTEP_BYTES, BSTEP_BYTES>::run(acc, a1, b1, a2, b2, aaddr, baddr);
    }
};

template <class Cfg, int ALAY, int BLAY, bool EDGE, class Epi>
__global__ void __launch_bounds__(Cfg::NTHREADS, Cfg::MINW) gemm_mfma_kernel(GemmProblem p, Epi epi) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, WM = Cfg::WM, WN = Cfg::WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NT = Cfg::NTHREADS;
    constexpr int MF = Cfg::MF;
    static_assert(BK % 8 == 0, "BK multiple of 8");
    typedef PanelGeom<ALAY, BM, BK, NT> GA;
    typedef PanelGeom<BLAY, BN, BK, NT> GB;

    // Tiles whose double-buffered panels exceed the 64 KiB static limit use dynamic LDS
    // (the launcher raises the function's dynamic-LDS cap once).
    constexpr int LDS_FLOATS = 2 * (GA::ELEMS + GB::ELEMS);
    constexpr bool DYN = (LDS_FLOATS * 4 > 65536);
    __shared__ __attribute__((aligned(16))) float smem_static[DYN ? 4 : LDS_FLOATS];
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    float* smem = DYN ? smem_dyn : smem_static;
    float* sA0 = smem;
    float* sB0 = smem + 2 * GA::ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, h = lane >> 5;

    // ---- tile decode (XCD aware) ------------------------------------------
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
    const int m0 = mt * BM;     // first OUTPUT row of this tile
    // A may be two row segments stacked (each with its own run of m tiles)
    const float* Ap = p.A;
    long lda = p.lda;
    int mA0 = m0, nArows = p.m_a1;
    if (mt >= p.tiles_m1) {
        Ap = p.A2;
        lda = p.lda2;
        mA0 = (mt - p.tiles_m1) * BM;
        nArows = p.M - p.m_a1;
    }
    const int kbeg = split * p.klen;
    const int kend = min(p.K, kbeg + p.klen);

    // B may be two column segments laid side by side ([Y | x] in the TN stats GEMM).
    // Each segment has its own run of n tiles, so a tile never straddles the seam.
    const float* Bp = p.B;
    long ldb = p.ldb;
    int nB0 = nt * BN;       // first row (= output column) of this tile inside its segment
    int nBrows = p.n_b1;     // rows of that segment
    int n0 = nB0;            // first output column
    int ncol_end = p.n_b1;   // output columns this tile may write: [n0, ncol_end)
    if (nt >= p.tiles_n1) {
        Bp = p.B2;
        ldb = p.ldb2;
        nB0 = (nt - p.tiles_n1) * BN;
        nBrows = p.N - p.n_b1;
        n0 = p.n_b1 + nB0;
        ncol_end = p.N;
    }

    // bounds-checked instantiation: may this tile's panels still use 16-byte loads? (see panel_gload)
    const bool a_al = EDGE && ((p.al_mask >> (mt >= p.tiles_m1 ? 1 : 0)) & 1) &&
                      (ALAY == KMAJOR || mA0 + BM <= nArows);
    const bool b_al = EDGE && ((p.al_mask >> (nt >= p.tiles_n1 ? 3 : 2)) & 1) &&
                      (BLAY == KMAJOR || nB0 + BN <= nBrows);

    if constexpr (MF == 16) {
        // ---- v_mfma_f32_16x16x4_f32 form: (WM/16) x (WN/16) accumulators of 4 registers ----
        static_assert(BK == 16 && Cfg::PIPE == 0, "16x16x4 form: BK = 16, plain schedule");
        static_assert(epi_mode<Epi>::value == 0 && !epi_rowbits<Epi>::value, "16x16x4 form: plain real epilogues only");
        constexpr int T16M = WM / 16, T16N = WN / 16;
        const int l15 = lane & 15, q = lane >> 4;
        f32x4 acc16[T16M][T16N];
#pragma unroll
        for (int i = 0; i < T16M; ++i)
#pragma unroll
            for (int j = 0; j < T16N; ++j) acc16[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 ra[GA::F4], rb[GB::F4];
        const int nkb = (kend - kbeg + BK - 1) / BK;
        if (nkb > 0) {
            panel_gload<ALAY, BM, BK, EDGE, NT>(ra, Ap, lda, mA0, nArows, kbeg, kend, tid, a_al && (kbeg) + BK <= kend);
            panel_gload<BLAY, BN, BK, EDGE, NT>(rb, Bp, ldb, nB0, nBrows, kbeg, kend, tid, b_al && (kbeg) + BK <= kend);
            panel_lds_store<ALAY, BM, BK, NT, MF>(sA0, ra, tid);
            panel_lds_store<BLAY, BN, BK, NT, MF>(sB0, rb, tid);
        }
        __syncthreads();
        for (int kb = 0; kb < nkb; ++kb) {
            const int cur = kb & 1;
            const float* sA = sA0 + cur * GA::ELEMS;
            const float* sB = sB0 + cur * GB::ELEMS;
            const bool more = (kb + 1) < nkb;
            if (more) {
                const int k0 = kbeg + (kb + 1) * BK;
                panel_gload<ALAY, BM, BK, EDGE, NT>(ra, Ap, lda, mA0, nArows, k0, kend, tid, a_al && (k0) + BK <= kend);
                panel_gload<BLAY, BN, BK, EDGE, NT>(rb, Bp, ldb, nB0, nBrows, k0, kend, tid, b_al && (k0) + BK <= kend);
            }
            f32x4 fa[T16M], fb[T16N];
#pragma unroll
            for (int i = 0; i < T16M; ++i) fa[i] = panel_frag16<ALAY, BM, BK>(sA, wm * WM + i * 16 + l15, q);
#pragma unroll
            for (int j = 0; j < T16N; ++j) fb[j] = panel_frag16<BLAY, BN, BK>(sB, wn * WN + j * 16 + l15, q);
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int i = 0; i < T16M; ++i)
#pragma unroll
                    for (int j = 0; j < T16N; ++j)
                        acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][st], fb[j][st], acc16[i][j],
                                                                           0, 0, 0);
            if (more) {
                panel_lds_store<ALAY, BM, BK, NT, MF>(sA0 + (cur ^ 1) * GA::ELEMS, ra, tid);
                panel_lds_store<BLAY, BN, BK, NT, MF>(sB0 + (cur ^ 1) * GB::ELEMS, rb, tid);
            }
            __syncthreads();
        }
        // C layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
        for (int i = 0; i < T16M; ++i)
#pragma unroll
            for (int j = 0; j < T16N; ++j) {
                const int col = n0 + wn * WN + j * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * WM + i * 16 + 4 * q + r;
                    if (!EDGE || (row < p.M && col < ncol_end)) epi(row, col, acc16[i][j][r], split);
                }
            }
        return;
    }

    constexpr bool ROWBITS = epi_rowbits<Epi>::value;
    uint32_t rbits[TM][TN];
    if constexpr (ROWBITS) {
        static_assert(epi_mode<Epi>::value == 0, "row-bit factors: real epilogues only");
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int brow = m0 + wm * WM + i * 32, bcol = n0 + wn * WN + j * 32 + l31;
                rbits[i][j] = (!EDGE || (brow < p.M && bcol < ncol_end)) ? epi.load_bits(brow, bcol) : 0u;
            }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 ra[GA::F4], rb[GB::F4];
    const int nkb = (kend - kbeg + BK - 1) / BK;

    if (nkb > 0) {
        panel_gload<ALAY, BM, BK, EDGE, NT>(ra, Ap, lda, mA0, nArows, kbeg, kend, tid, a_al && (kbeg) + BK <= kend);
        panel_gload<BLAY, BN, BK, EDGE, NT>(rb, Bp, ldb, nB0, nBrows, kbeg, kend, tid, b_al && (kbeg) + BK <= kend);
        panel_lds_store<ALAY, BM, BK, NT, MF>(sA0, ra, tid);
        panel_lds_store<BLAY, BN, BK, NT, MF>(sB0, rb, tid);
    }
    __syncthreads();

    if constexpr (Cfg::PIPE == 2) {
        // All fragment reads of a K block are issued right after the barrier, ahead of its first
        // MFMA (the compiler's own schedule re-uses one register set and reads group c + 1 only
        // when group c has been issued: with the four waves of a SIMD phase-locked by the barrier
        // that leaves the matrix pipe idle for one LDS round trip per group).
        constexpr int NC = BK / 8;
        for (int kb = 0; kb < nkb; ++kb) {
            const int cur = kb & 1;
            const float* sA = sA0 + cur * GA::ELEMS;
            const float* sB = sB0 + cur * GB::ELEMS;
            const bool more = (kb + 1) < nkb;
            f32x4 fa[NC][TM], fb[NC][TN];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[c][i] = panel_frag<ALAY, BM, BK>(sA, wm * WM + i * 32 + l31, c, h);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[c][j] = panel_frag<BLAY, BN, BK>(sB, wn * WN + j * 32 + l31, c, h);
            }
            if (more) {  // next block's global loads, behind the LDS reads in issue order
                const int k0 = kbeg + (kb + 1) * BK;
                panel_gload<ALAY, BM, BK, EDGE, NT>(ra, Ap, lda, mA0, nArows, k0, kend, tid, a_al && (k0) + BK <= kend);
                panel_gload<BLAY, BN, BK, EDGE, NT>(rb, Bp, ldb, nB0, nBrows, k0, kend, tid, b_al && (k0) + BK <= kend);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i][s], fb[c][j][s],
                                                                             acc[i][j], 0, 0, 0);
            if (more) {
                panel_lds_store<ALAY, BM, BK, NT, MF>(sA0 + (cur ^ 1) * GA::ELEMS, ra, tid);
                panel_lds_store<BLAY, BN, BK, NT, MF>(sB0 + (cur ^ 1) * GB::ELEMS, rb, tid);
            }
            __syncthreads();
        }
    } else if constexpr (Cfg::PIPE == 3) {
        static_assert(ALAY == XMAJOR && BLAY == XMAJOR && TM == 2 && TN == 2, "pair schedule: row-contiguous panels, 64 x 64 wave tiles");
        static_assert(!ROWBITS, "pair schedule: no row-bit epilogues");
        constexpr int NS = BK / 2;
        for (int kb = 0; kb < nkb; ++kb) {
            const int cur = kb & 1;
            const unsigned aaddr = lds_addr_of(sA0 + cur * GA::ELEMS + h * GA::STRIDE + wm * WM + 2 * l31);
            const unsigned baddr = lds_addr_of(sB0 + cur * GB::ELEMS + h * GB::STRIDE + wn * WN + 2 * l31);
            const bool more = (kb + 1) < nkb;
            if (more) {
                const int k0 = kbeg + (kb + 1) * BK;
                panel_gload<ALAY, BM, BK, EDGE, NT>(ra, Ap, lda, mA0, nArows, k0, kend, tid, a_al && (k0) + BK <= kend);
                panel_gload<BLAY, BN, BK, EDGE, NT>(rb, Bp, ldb, nB0, nBrows, k0, kend, tid, b_al && (k0) + BK <= kend);
            }
            pair_f32x2 a0, b0, a1, b1;
            lds_read_b64_at<0>(a0, aaddr);
            lds_read_b64_at<0>(b0, baddr);
            lds_read_b64_at<2 * GA::STRIDE * 4>(a1, aaddr);
            lds_read_b64_at<2 * GB::STRIDE * 4>(b1, baddr);
            PairSteps<0, NS, 2 * GA::STRIDE * 4, 2 * GB::STRIDE * 4>::run(acc, a0, b0, a1, b1, aaddr, baddr);
            if (more) {
                panel_lds_store<ALAY, BM, BK, NT, MF>(sA0 + (cur ^ 1) * GA::ELEMS, ra, tid);
                panel_lds_store<BLAY, BN, BK, NT, MF>(sB0 + (cur ^ 1) * GB::ELEMS, rb, tid);
            }
            __syncthreads();
        }
    } else if constexpr (Cfg::PIPE == 0) {
        for (int kb = 0; kb < nkb; ++kb) {
            const int cur = kb & 1;
            const float* sA = sA0 + cur * GA::ELEMS;
            const float* sB = sB0 + cur * GB::ELEMS;
            const bool more = (kb + 1) < nkb;
            if (more) {  // issue next block's global loads; they land during the MFMAs
                const int k0 = kbeg + (kb + 1) * BK;
                panel_gload<ALAY, BM, BK, EDGE, NT>(ra, Ap, lda, mA0, nArows, k0, kend, tid, a_al && (k0) + BK <= kend);
                panel_gload<BLAY, BN, BK, EDGE, NT>(rb, Bp, ldb, nB0, nBrows, k0, kend, tid, b_al && (k0) + BK <= kend);
            }
    #pragma unroll
            for (int c = 0; c < BK / 8; ++c) {
                f32x4 fa[TM], fb[TN];
    #pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = panel_frag<ALAY, BM, BK>(sA, wm * WM + i * 32 + l31, c, h);
    #pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = panel_frag<BLAY, BN, BK>(sB, wn * WN + j * 32 + l31, c, h);
    #pragma unroll
                for (int s = 0; s < 4; ++s)
    #pragma unroll
                    for (int i = 0; i < TM; ++i)
    #pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s],
                                                                             acc[i][j], 0, 0, 0);
            }
            if (more) {
                panel_lds_store<ALAY, BM, BK, NT, MF>(sA0 + (cur ^ 1) * GA::ELEMS, ra, tid);
                panel_lds_store<BLAY, BN, BK, NT, MF>(sB0 + (cur ^ 1) * GB::ELEMS, rb, tid);
            }
            __syncthreads();
        }
    } else {
        constexpr int NC = BK / 8;
        static_assert(NC >= 2, "PIPE needs at least two MFMA groups per K block");
        // Schedule per K block (registers ra/rb always hold the block AFTER the one in LDS):
        //   groups 0 .. NC-2 : read the next group's operands, MFMA
        //   last group       : publish block kb+1 (ra/rb -> other LDS buffer), barrier, start the
        //                      global loads of block kb+2 (a whole K block of MFMAs to land),
        //                      prefetch group 0 of block kb+1, MFMA with operands already held
        f32x4 fa[TM], fb[TN];
        if (nkb > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = panel_frag<ALAY, BM, BK>(sA0, wm * WM + i * 32 + l31, 0, h);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = panel_frag<BLAY, BN, BK>(sB0, wn * WN + j * 32 + l31, 0, h);
        }
        if (nkb > 1) {
            panel_gload<ALAY, BM, BK, EDGE, NT>(ra, Ap, lda, mA0, nArows, kbeg + BK, kend, tid, a_al && (kbeg + BK) + BK <= kend);
            panel_gload<BLAY, BN, BK, EDGE, NT>(rb, Bp, ldb, nB0, nBrows, kbeg + BK, kend, tid, b_al && (kbeg + BK) + BK <= kend);
        }
        for (int kb = 0; kb < nkb; ++kb) {
            const int cur = kb & 1;
            const float* sA = sA0 + cur * GA::ELEMS;
            const float* sB = sB0 + cur * GB::ELEMS;
            const bool more = (kb + 1) < nkb;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f32x4 na[TM], nb[TN];
                if (c + 1 < NC) {          // operands of the next group of THIS block
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        na[i] = panel_frag<ALAY, BM, BK>(sA, wm * WM + i * 32 + l31, c + 1, h);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        nb[j] = panel_frag<BLAY, BN, BK>(sB, wn * WN + j * 32 + l31, c + 1, h);
                } else {
                    if (more) {
                        panel_lds_store<ALAY, BM, BK, NT, MF>(sA0 + (cur ^ 1) * GA::ELEMS, ra, tid);
                        panel_lds_store<BLAY, BN, BK, NT, MF>(sB0 + (cur ^ 1) * GB::ELEMS, rb, tid);
                    }
                    __syncthreads();
                    if ((kb + 2) < nkb) {
                        const int k0 = kbeg + (kb + 2) * BK;
                        panel_gload<ALAY, BM, BK, EDGE, NT>(ra, Ap, lda, mA0, nArows, k0, kend, tid, a_al && (k0) + BK <= kend);
                        panel_gload<BLAY, BN, BK, EDGE, NT>(rb, Bp, ldb, nB0, nBrows, k0, kend, tid, b_al && (k0) + BK <= kend);
                    }
                    if (more) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            na[i] = panel_frag<ALAY, BM, BK>(sA0 + (cur ^ 1) * GA::ELEMS,
                                                             wm * WM + i * 32 + l31, 0, h);
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            nb[j] = panel_frag<BLAY, BN, BK>(sB0 + (cur ^ 1) * GB::ELEMS,
                                                             wn * WN + j * 32 + l31, 0, h);
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s],
                                                                             acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = na[i];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = nb[j];
            }
        }
    }

    // ---- epilogue: C layout of the 32x32 MFMA: col = lane&31,
    //      row = (r&3) + 8*(r>>2) + 4*(lane>>5) ------------------------------
    // Epi::kMode (optional member) selects how accumulators reach the functor:
    //   0  one real value per element                          epi(row, col, v, split)
    //   1  complex product on real "extended" operands, output columns interleave (re, im):
    //      lanes 2q / 2q+1 hold the two parts -> one shuffle, even lanes call the complex functor
    //   2  complex x^H y on the real views of both operands: the 2x2 block of real sums
    //      (rr ri; ir ii) sits in registers r, r+1 of lanes 2q, 2q+1:
    //      re = rr + ii, im = ri - ir  (two shuffles), even lanes call the complex functor
    //   3  complex x y with the planar rows of x against the real view of y: the same 2x2 block,
    //      re = rr - ii, im = ri + ir
    constexpr int MODE = epi_mode<Epi>::value;
    if constexpr (Cfg::PIPE == 3) {
        // pair schedule: accumulator (i, j) register r of lane (l31, h) is output row 2 rin + i, column 2 l31 + j
        static_assert(MODE == 0 || MODE == 2, "pair schedule: real outputs or the complex x^H y combination");
        const int rbase = m0 + wm * WM, cbase = n0 + wn * WN + 2 * l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
            if constexpr (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = rbase + 2 * rin + i;
                    const float v0 = acc[i][0][r], v1 = acc[i][1][r];
                    if (!EDGE || (row < p.M && cbase + 1 < ncol_end)) {
                        if constexpr (epi_vec2<Epi>::value) {
                            if (p.vec_epi) epi.store2(row, cbase, v0, v1, split);
                            else { epi(row, cbase, v0, split); epi(row, cbase + 1, v1, split); }
                        } else {
                            epi(row, cbase, v0, split);
                            epi(row, cbase + 1, v1, split);
                        }
                    } else if (row < p.M && cbase < ncol_end) {
                        epi(row, cbase, v0, split);
                    }
                }
            } else {
                // rows 2 m, 2 m + 1 = (re, im) of x column m, columns 2 n, 2 n + 1 = (re, im) of y column n:
                // (rr ri; ir ii) = (acc00 acc01; acc10 acc11);  x^H y: re = rr + ii, im = ri - ir
                const float re = acc[0][0][r] + acc[1][1][r];
                const float im = acc[0][1][r] - acc[1][0][r];
                const int row = rbase + 2 * rin;
                if (!EDGE || (row + 1 < p.M && cbase + 1 < ncol_end)) epi.pair(row >> 1, cbase >> 1, re, im, split);
            }
        }
        return;
    }
    if constexpr ((MODE == 0 || MODE == 1) && !EDGE && !ROWBITS && epi_vec4<Epi>::value) {
        if (p.vec_epi) {
            const int t = l31 & 3, col0 = (l31 & ~3);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 x;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            x[e] = acc[i][j][4 * g + e];
                        }
                        const f32x4 y = quad_transpose(x, t);
                        epi.vec4(m0 + wm * WM + i * 32 + 8 * g + 4 * h + t, n0 + wn * WN + j * 32 + col0, y, split);
                    }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + l31;
            if constexpr (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int row = m0 + wm * WM + i * 32 + rin;
                    if (!EDGE || (row < p.M && col < ncol_end)) {
                        if constexpr (ROWBITS) epi.with_bit(row, col, acc[i][j][r], (rbits[i][j] >> rin) & 1u, split);
                        else epi(row, col, acc[i][j][r], split);
                    }
                }
            } else if constexpr (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float v = acc[i][j][r];
                    const float o = __shfl_xor(v, 1, 64);   // executed by every lane
                    if (!(lane & 1) && (!EDGE || (row < p.M && col < ncol_end)))
                        epi.pair(row, col >> 1, v, o, split);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float v0 = acc[i][j][r], v1 = acc[i][j][r + 1];
                    const float p1 = __shfl_xor(v1, 1, 64);
                    // mode 2  odd: ri - ir, even: rr + ii ;  mode 3  odd: ri + ir, even: rr - ii
                    const bool plus = (MODE == 2) ? !(lane & 1) : (lane & 1);
                    const float comb = plus ? (v0 + p1) : (v0 - p1);
                    const float other = __shfl_xor(comb, 1, 64);
                    if (!(lane & 1) && (!EDGE || (row < p.M && col < ncol_end)))
                        epi.pair(row >> 1, col >> 1, comb, other, split);
                }
            }
        }
    }
}

// hipFuncSetAttribute is per device: remember per (kernel instantiation, device) whether the
// dynamic-LDS cap has been raised (one process normally drives one GPU, but nothing here assumes it).
// (atomic flags: handles of different host threads may launch the same instantiation; raising the cap twice
//  is harmless, a torn flag is not)
struct DynLdsRaised {
    std::atomic<bool> done[32];
    DynLdsRaised() {
        for (auto& d : done) d.store(false, std::memory_order_relaxed);
    }
    std::atomic<bool>& on_current_device() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        return done[(dev >= 0 && dev < 32) ? dev : 0];
    }
};

// Host-side launch helper.  Picks EDGE when any fast-path precondition fails.
template <class Cfg, int ALAY, int BLAY, class Epi>
inline hipError_t launch_gemm_mfma(hipStream_t stream, GemmProblem p, const Epi& epi) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    if (p.B2 == nullptr) p.n_b1 = p.N;
    if (p.A2 == nullptr) p.m_a1 = p.M;
    // Stacked A: the second segment's tiles start at OUTPUT row tiles_m1 * BM, which is m_a1 only when the
    // first segment is a whole number of tiles (the bounds-checked instantiation does not re-map rows).
    if (p.A2 != nullptr && (p.m_a1 % BM) != 0) return hipErrorInvalidValue;
    p.tiles_m1 = (p.m_a1 + BM - 1) / BM;
    p.tiles_m = p.tiles_m1 + (p.M - p.m_a1 + BM - 1) / BM;
    p.tiles_n1 = (p.n_b1 + BN - 1) / BN;
    p.tiles_n = p.tiles_n1 + (p.N - p.n_b1 + BN - 1) / BN;
    if (p.ksplits < 1) p.ksplits = 1;
    if (p.ksplits == 1) {
        p.klen = ((p.K + BK - 1) / BK) * BK;
        if (p.klen == 0) p.klen = BK;
    }
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    // every split range [s*klen, min(K, (s+1)*klen)) is then a whole number of K blocks
    bool fast = (p.M % BM == 0) && (p.N % BN == 0) && (p.K % BK == 0) && (p.klen % BK == 0) &&
                (p.lda % 4 == 0) && (p.ldb % 4 == 0) && al16(p.A) && al16(p.B) && p.K > 0;
    if (p.B2 != nullptr) fast = fast && (p.n_b1 % BN == 0) && (p.ldb2 % 4 == 0) && al16(p.B2);
    if (p.A2 != nullptr) fast = fast && (p.m_a1 % BM == 0) && (p.lda2 % 4 == 0) && al16(p.A2);
    p.vec_epi = (fast && epi_vec_ok(epi)) ? 1 : 0;
    if constexpr (Cfg::PIPE == 3) {
        // pair schedule: vec_epi = the functor's 8-byte store of two adjacent columns may be used (every
        // column origin of a lane is even: segment seams and leading dims even, base 8-byte aligned)
        p.vec_epi = (epi_vec2_ok(epi) && (p.n_b1 % 2) == 0) ? 1 : 0;
    }
    p.al_mask = ((al16(p.A) && p.lda % 4 == 0) ? 1 : 0) | ((p.A2 != nullptr && al16(p.A2) && p.lda2 % 4 == 0) ? 2 : 0) |
                ((al16(p.B) && p.ldb % 4 == 0) ? 4 : 0) | ((p.B2 != nullptr && al16(p.B2) && p.ldb2 % 4 == 0) ? 8 : 0);
    const int grid = p.tiles_m * p.tiles_n * p.ksplits;
    if (grid <= 0) return hipSuccess;
    constexpr int lds_bytes = 8 * (PanelGeom<ALAY, BM, BK, Cfg::NTHREADS>::ELEMS +
                                   PanelGeom<BLAY, BN, BK, Cfg::NTHREADS>::ELEMS);
    constexpr int dyn_bytes = lds_bytes > 65536 ? lds_bytes : 0;
    if (dyn_bytes) {
        static DynLdsRaised raised_fast, raised_edge;   // per instantiation
        std::atomic<bool>& raised = fast ? raised_fast.on_current_device() : raised_edge.on_current_device();
        if (!raised) {
            const void* fn = fast ? reinterpret_cast<const void*>(&gemm_mfma_kernel<Cfg, ALAY, BLAY, false, Epi>)
                                  : reinterpret_cast<const void*>(&gemm_mfma_kernel<Cfg, ALAY, BLAY, true, Epi>);
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, dyn_bytes);
            if (e != hipSuccess) return e;
            raised = true;
        }
    }
    if (fast)
        hipLaunchKernelGGL((gemm_mfma_kernel<Cfg, ALAY, BLAY, false, Epi>), dim3(grid), dim3(Cfg::NTHREADS),
                           dyn_bytes, stream, p, epi);
    else
        hipLaunchKernelGGL((gemm_mfma_kernel<Cfg, ALAY, BLAY, true, Epi>), dim3(grid), dim3(Cfg::NTHREADS),
                           dyn_bytes, stream, p, epi);
    return hipGetLastError();
}

}  // namespace dcp
