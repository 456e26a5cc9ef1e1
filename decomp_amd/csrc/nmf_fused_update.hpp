// The replicated D side of one MU iteration (l2, no mask, float32) as ONE plain launch.
//
// Reference semantics (decomp/nmf_methods/batch_mu.py:18-24, grads.py:86-93, 117-125,
// utils/normalize.py:13-21), in the Gram formulation of nmf_impl.hpp:
//   S = x^T x, num = x^T Y           (statistics; here: split-K partials or the all-reduced array)
//   U = D o max(num, 0) / max(S D, 1e-15)
//   D_new = U / |U|_rows ; max|D - D_new| ; G = D_new D_new^T  (the next iteration's x gradient needs it)
// Before this kernel the chain was five latency-bound launches (slab sum, S.D + quotient, row normalise,
// Gram split-K, its slab sum) plus a 4-byte copy: ~55 us of kernel time and ~8 dependent kernel boundaries
// for < 10 us of work.  Here the phases are separated by grid barriers (grid_barrier.hpp):
//   A  S <- ordered sum of the statistics slabs' x^T x block             (single-GPU form only)
//   -- barrier --
//   B  per 64 x 64 tile of [K, F]: den = S D on the fp32 MFMA, U = D o max(num, 0) / max(den, eps) (num summed
//      from the slabs in order), U stored, per-row partial sums of U^2
//   -- barrier --
//   C  row norms from the partials (fixed order), D_new = U / norm, per-workgroup max|D - D_new|
//   -- barrier --
//   D  G partials: 64 x 64 tiles of D_new D_new^T over F slices (MFMA)
//   -- barrier --
//   E  G <- ordered sum of the partials; block 0: max over workgroups -> maxdiff
// Every sum has a fixed order: results are bitwise reproducible run to run.  The slab sums use the same
// left-to-right order as reduce_slabs_kernel, the quotient the same expression as EpiMuDen.
#pragma once
#include <math.h>

#include "gemm_mfma_f32.hpp"
#include "grid_barrier.hpp"
#include "handle.hpp"
#include "kernels_small.hpp"
#include "scalar.hpp"

namespace dcp {

struct FusedUpdArgs {
    // statistics [K, W], W = F + K: columns [0, F) = x^T Y, [F, F + K) = x^T x.
    // nslabs >= 1: split-K partials, element (s, r, c) at stats[s * slab_stride + r * W + c]
    // nslabs == 0: the reduced array
    const float* stats;
    long slab_stride;
    int nslabs;
    int W;
    const float* D;
    float* D_new;
    float* U;         // [K, F] scratch
    float* Sred;      // [K, K] scratch (nslabs >= 1)
    float* rowpart;   // [K][tiles_f]
    float* gslabs;    // [gram_slices][K * K]
    float* G;         // [K, K] out: D_new D_new^T
    float* wgmax;     // [gridDim.x]
    float* maxdiff_out;    // device or pinned host memory
    float* maxdiff_zero;   // nullable: cleared (the ping-pong slot of the step API)
    int* status_out;       // nullable: 1 when a barrier spin expired
    GridBarrierState* bar;
    int K, F;
    int tiles_k, tiles_f;
    int gram_slices, slice_cols;
    // analysis only (DCP_FUSED_STAMPS=1): s_memrealtime (100 MHz) at the phase boundaries, written by lane 0
    // of blocks 0 and gridDim.x - 1: stamps[block_sel * 16 + i]
    unsigned long long* stamps;
};

__device__ __forceinline__ void fu_stamp(const FusedUpdArgs& a, int i) {
    if (a.stamps != nullptr && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))
        a.stamps[(blockIdx.x == 0 ? 0 : 16) + i] = __builtin_amdgcn_s_memrealtime();
}

constexpr int kFuLd = 68;   // LDS row stride in floats (16-byte aligned rows)

// acc += A[arow0 + r, k] * B(k, n) for the 32 x 32 sub-tile of wave (wm, wn) of a 64 x 64 tile, k in [k0, k1).
// A: row-major [rows, lda] (reduction index contiguous).  B: BLAY == KMAJOR: row-major [n, ldb] with the
// reduction index contiguous (A . B^T); BLAY == XMAJOR: row-major [k, ldb] (A . B).  Rows / columns past the
// valid counts and k >= k1 contribute zeros.  All leading dims, k0, k1 and column origins are multiples of 4
// and the bases 16-byte aligned (checked by the launcher).
// The reduction runs in groups of 128 (two 64-deep LDS chunks): a group's 16 panel loads per thread are
// issued together -- this kernel has ONE 4-wave workgroup per CU, so what hides memory latency is the number
// of loads each wave keeps in flight, not other waves -- and the next group's loads are issued before the
// current group's MFMAs.
template <int BLAY>
__device__ __forceinline__ void fu_mma64(f32x16& acc, const float* __restrict__ A, long lda, int arow0,
                                         int arows, const float* __restrict__ B, long ldb, int b0, int bvalid,
                                         int k0, int k1, float* sA, float* sB, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    constexpr int NCH = 2;       // 64-deep chunks per group (registers: 8 NCH float4 per thread)
    f32x4 ra[NCH][4], rb[NCH][4];
    auto gload = [&](int kg) {   // chunks kg, kg + 64, ...
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int kc = kg + 64 * c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 4, kq = (idx & 15) << 2;
                {   // A chunk: 64 rows x 64 k
                    const int row = arow0 + r;
                    const bool ok = (row < arows) && (kc + kq < k1);
                    const float* p = A + (long)(ok ? row : 0) * lda + (ok ? kc + kq : 0);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
                    ra[c][i] = ok ? v : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
                if (BLAY == KMAJOR) {   // B chunk: 64 n-rows x 64 k
                    const int row = b0 + r;
                    const bool ok = (row < bvalid) && (kc + kq < k1);
                    const float* p = B + (long)(ok ? row : 0) * ldb + (ok ? kc + kq : 0);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
                    rb[c][i] = ok ? v : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                } else {                // B chunk: 64 k-rows x 64 n
                    const int kr = kc + r, col = b0 + kq;
                    const bool ok = (kr < k1) && (col < bvalid);
                    const float* p = B + (long)(ok ? kr : 0) * ldb + (ok ? col : 0);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
                    rb[c][i] = ok ? v : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
        }
    };
    if (k0 >= k1) return;
    gload(k0);
    for (int kg = k0; kg < k1; kg += 64 * NCH) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (kg + 64 * c < k1) {     // (wave-uniform)
                __syncthreads();        // the previous chunk's fragment reads are done
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = tid + 256 * i;
                    const int r = idx >> 4, kq = (idx & 15) << 2;
                    *reinterpret_cast<f32x4*>(sA + r * kFuLd + kq) = ra[c][i];
                    *reinterpret_cast<f32x4*>(sB + r * kFuLd + kq) = rb[c][i];
                }
                __syncthreads();
                if (c == NCH - 1 && kg + 64 * NCH < k1) gload(kg + 64 * NCH);   // lands under the MFMAs below
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 fa = *reinterpret_cast<const f32x4*>(sA + (wm * 32 + l31) * kFuLd + 8 * g + 4 * h);
                    f32x4 fb;
                    if (BLAY == KMAJOR) {
                        fb = *reinterpret_cast<const f32x4*>(sB + (wn * 32 + l31) * kFuLd + 8 * g + 4 * h);
                    } else {
                        const float* q = sB + (8 * g + 4 * h) * kFuLd + wn * 32 + l31;
                        fb[0] = q[0];
                        fb[1] = q[kFuLd];
                        fb[2] = q[2 * kFuLd];
                        fb[3] = q[3 * kFuLd];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc, 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();
}

// the 32 x 32 accumulator of wave (wm, wn) -> the 64 x 64 row-major LDS tile
__device__ __forceinline__ void fu_acc_to_lds(const f32x16& acc, float* sU, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        sU[row * kFuLd + wn * 32 + l31] = acc[r];
    }
}

// acc[i] = ordered sum over the S slabs of the 4 consecutive elements at offset o[i] of a slab, i < NV, the
// loads of eight slabs (8 NV of them) in flight together.  Same left-to-right order per element as
// reduce_slabs_kernel.
template <int NV>
__device__ __forceinline__ void fu_slab_sum(f32x4 (&acc)[NV], const float* __restrict__ base, const long (&o)[NV],
                                            long stride, int S) {
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = *reinterpret_cast<const f32x4*>(base + o[i]);
    int s = 1;
    for (; s + 7 < S; s += 8) {
        f32x4 v[8][NV];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i) v[u][i] = *reinterpret_cast<const f32x4*>(base + (long)(s + u) * stride + o[i]);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][e] = acc[i][e] + v[u][i][e];
    }
    for (; s < S; ++s) {
        f32x4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(base + (long)s * stride + o[i]);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] = acc[i][e] + v[i][e];
    }
}

// (a template only so that the header may be included by several translation units)
// __launch_bounds__(256, 2): at most 256 registers per lane, so that TWO such workgroups fit a CU -- a second
// process driving the same GPU (two ranks on one device in the tests) can then always become resident beside
// this launch instead of dead-locking both grids.
template <int VARIANT>
__global__ void __launch_bounds__(256, 2) nmf_fused_update_kernel(FusedUpdArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[64 * kFuLd];
    __shared__ __attribute__((aligned(16))) float sB[64 * kFuLd];
    __shared__ __attribute__((aligned(16))) float sU[64 * kFuLd];
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    const int K = a.K, F = a.F, W = a.W;
    const long KK = (long)K * K;
    const int ntiles = a.tiles_k * a.tiles_f;
    const int row_t = tid >> 2, q = tid & 3;   // epilogue thread map: row of the tile, 16-column quarter
    unsigned epoch = 0;
    // everything another workgroup reads later in this launch is stored write-through (sc1): no L2
    // write-back is needed before a barrier arrival (grid_barrier<true>)
    const __amdgpu_buffer_rsrc_t rS = gb_rsrc(a.Sred), rU = gb_rsrc(a.U), rP = gb_rsrc(a.rowpart),
                                 rD = gb_rsrc(a.D_new), rGs = gb_rsrc(a.gslabs), rM = gb_rsrc(a.wgmax);
    fu_stamp(a, 0);

    // the num (x^T Y) quarter-rows of a tile: 4 float4 per thread, summed over the slabs in order
    auto load_num = [&](int tile, f32x4 (&nu)[4]) {
        const int ti = tile % a.tiles_k, tf = tile / a.tiles_k;
        const int grow = ti * 64 + row_t;
        long o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gcol = tf * 64 + q * 16 + 4 * i;
            const bool ok = (grow < K) && (gcol < F);
            o[i] = (long)(ok ? grow : 0) * W + (ok ? gcol : 0);
        }
        fu_slab_sum<4>(nu, a.stats, o, a.slab_stride, a.nslabs >= 1 ? a.nslabs : 1);
    };

    // ---- A: this workgroup's first num tile (independent of S: its loads overlap the barrier), and
    //         S = sum of the slabs' x^T x block (one wave per workgroup takes a share) -------------------------
    f32x4 nu0[4];
    if ((int)blockIdx.x < ntiles) load_num(blockIdx.x, nu0);
    const float* S = a.stats + F;   // reduced statistics: S at column F, leading dim W
    long ldS = W;
    if (a.nslabs >= 1) {
        if (tid < 64) {
            for (long i4 = (long)blockIdx.x * 64 + tid; i4 < KK / 4; i4 += (long)gridDim.x * 64) {
                const long e = i4 * 4;
                const long row = e / K, col = e - row * K;
                f32x4 v[1];
                const long o[1] = {row * W + F + col};
                fu_slab_sum<1>(v, a.stats, o, a.slab_stride, a.nslabs);
                st_sc1_f4(rS, (unsigned)(e * 4), v[0]);
            }
        }
        S = a.Sred;
        ldS = K;
        fu_stamp(a, 1);
        grid_barrier<true>(a.bar, ++epoch);
    }
    fu_stamp(a, 2);

    // ---- B: U tile = D o max(num, 0) / max(S D, eps), row partial sums of U^2 ---------------------
    // The first tile's U stays in registers across the barrier; a workgroup that owns more tiles (more tiles
    // than resident workgroups) parks the others in the U scratch.
    f32x4 u0[4];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const bool first = (tile == (int)blockIdx.x);
        const int ti = tile % a.tiles_k, tf = tile / a.tiles_k;
        f32x4 nu[4];
        if (first) {
#pragma unroll
            for (int i = 0; i < 4; ++i) nu[i] = nu0[i];
        } else {
            load_num(tile, nu);
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        fu_mma64<XMAJOR>(acc, S, ldS, ti * 64, K, a.D, (long)F, tf * 64, F, 0, K, sA, sB, tid);
        fu_acc_to_lds(acc, sU, tid);
        __syncthreads();
        const int grow = ti * 64 + row_t;
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gcol = tf * 64 + q * 16 + 4 * i;
            const bool ok = (grow < K) && (gcol < F);
            const f32x4 d = *reinterpret_cast<const f32x4*>(a.D + (long)(ok ? grow : 0) * F + (ok ? gcol : 0));
            const f32x4 den = *reinterpret_cast<const f32x4*>(sU + row_t * kFuLd + q * 16 + 4 * i);
            f32x4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u[e] = d[e] * max_np(nu[i][e], 0.0f) / max_np(den[e], 1.0e-15f);   // grads.py:93
                if (!ok) u[e] = 0.0f;
                ss += u[e] * u[e];
            }
            if (first) u0[i] = u;
            else if (ok) st_sc1_f4(rU, (unsigned)(((long)grow * F + gcol) * 4), u);
        }
        ss = ss + __shfl_xor(ss, 1, 64);   // (q0 + q1), (q2 + q3)
        ss = ss + __shfl_xor(ss, 2, 64);   // ((q0 + q1) + (q2 + q3)) on every lane of the quad
        if (q == 0 && grow < K) st_sc1_f1(rP, (unsigned)(((long)grow * a.tiles_f + tf) * 4), ss);
        __syncthreads();   // sU is rewritten by the next tile
    }
    fu_stamp(a, 3);
    grid_barrier<true>(a.bar, ++epoch);
    fu_stamp(a, 4);

    // ---- C: D_new = U / |U|, max|D - D_new| ---------------------------------------------------------------
    float md = 0.0f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const bool first = (tile == (int)blockIdx.x);
        const int ti = tile % a.tiles_k, tf = tile / a.tiles_k;
        const int grow = ti * 64 + row_t;
        const bool rok = grow < K;
        float tot = 0.0f;
        {   // this row's partials, strided over the quad, 8 loads in flight
            const float* rp = a.rowpart + (long)(rok ? grow : 0) * a.tiles_f;
            int j = q;
            for (; j + 28 < a.tiles_f; j += 32) {
                float v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = rp[j + 4 * t];
#pragma unroll
                for (int t = 0; t < 8; ++t) tot += v[t];
            }
            for (; j < a.tiles_f; j += 4) tot += rp[j];
        }
        tot = tot + __shfl_xor(tot, 1, 64);
        tot = tot + __shfl_xor(tot, 2, 64);
        const float nrm = sqrtf(tot);           // normalize.py:13-21 (l2_strict)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gcol = tf * 64 + q * 16 + 4 * i;
            if (rok && gcol < F) {
                const long o = (long)grow * F + gcol;
                const f32x4 u = first ? u0[i] : *reinterpret_cast<const f32x4*>(a.U + o);
                const f32x4 d = *reinterpret_cast<const f32x4*>(a.D + o);
                f32x4 dn;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dn[e] = u[e] / nrm;                         // a true division, as the reference's U / sqrt(.)
                    const float df = fabsf(d[e] - dn[e]);
                    md = (df > md || df != df) ? df : md;       // NaN propagates, as np.max
                }
                st_sc1_f4(rD, (unsigned)(o * 4), dn);
            }
        }
    }
    {
        const float m = block_max_256(md, s_red);
        if (tid == 0) st_sc1_f1(rM, (unsigned)(blockIdx.x * 4), m);
    }
    fu_stamp(a, 5);
    grid_barrier<true>(a.bar, ++epoch);
    fu_stamp(a, 6);

    // ---- D: partials of G = D_new D_new^T over F slices ---------------------------------------------------
    const int gtiles = a.tiles_k * a.tiles_k;
    const int gunits = gtiles * a.gram_slices;
    for (int unit = blockIdx.x; unit < gunits; unit += gridDim.x) {
        const int slice = unit / gtiles, tt = unit - slice * gtiles;
        const int ti = tt % a.tiles_k, tj = tt / a.tiles_k;
        const int f0 = slice * a.slice_cols;
        const int f1 = min(F, f0 + a.slice_cols);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        fu_mma64<KMAJOR>(acc, a.D_new, (long)F, ti * 64, K, a.D_new, (long)F, tj * 64, K, f0, f1, sA, sB, tid);
        fu_acc_to_lds(acc, sU, tid);
        __syncthreads();
        const int grow = ti * 64 + row_t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gcol = tj * 64 + q * 16 + 4 * i;
            if (grow < K && gcol < K)
                st_sc1_f4(rGs, (unsigned)(((long)slice * KK + (long)grow * K + gcol) * 4),
                          *reinterpret_cast<const f32x4*>(sU + row_t * kFuLd + q * 16 + 4 * i));
        }
        __syncthreads();
    }
    fu_stamp(a, 7);
    grid_barrier<true>(a.bar, ++epoch);
    fu_stamp(a, 8);

    // ---- E: max|D - D_new| over the workgroups (last block first: it has the least to sum);
    //         G = ordered sum of the partials -------------------------------------------------------------
    if (blockIdx.x == gridDim.x - 1) {
        float m = 0.0f;
        for (int i = tid; i < (int)gridDim.x; i += 256) {
            const float v = a.wgmax[i];
            m = (v > m || v != v) ? v : m;
        }
        m = block_max_256(m, s_red);
        if (tid == 0) {
            const unsigned expired = __hip_atomic_load(&a.bar->timeout.v, DCP_RLX_AGENT);
            *a.maxdiff_out = expired ? __builtin_nanf("") : m;
            if (a.maxdiff_zero != nullptr) *a.maxdiff_zero = 0.0f;
            if (a.status_out != nullptr) *a.status_out = expired ? 1 : 0;
        }
    }
    if (tid < 64) {
        for (long i4 = (long)blockIdx.x * 64 + tid; i4 < KK / 4; i4 += (long)gridDim.x * 64) {
            f32x4 v[1];
            const long o[1] = {i4 * 4};
            fu_slab_sum<1>(v, a.gslabs, o, KK, a.gram_slices);
            *reinterpret_cast<f32x4*>(a.G + i4 * 4) = v[0];
        }
    }
    fu_stamp(a, 9);
    grid_barrier_finish(a.bar);
    fu_stamp(a, 10);
}

// ---- host side ---------------------------------------------------------------------------------------------
struct FusedUpdPlan {
    int grid = 0;
    int tiles_k = 0, tiles_f = 0, gram_slices = 0, slice_cols = 0;
    size_t U_elems = 0, rowpart_elems = 0, gslab_elems = 0, wgmax_elems = 0;
};

// May the fused launch serve this update?  (float32, l2 without mask; float4 paths need K, F multiples of 4.)
inline bool fused_update_usable(int64_t F, int64_t K) {
    const bool off = getenv("DCP_NO_FUSED_UPDATE") != nullptr;   // A/B and test knob, read per call
    // (byte offsets of the write-through stores are 32-bit: every exchanged array stays below 2 GiB)
    return !off && (K % 4 == 0) && (F % 4 == 0) && K >= 4 && F >= 4 && K <= 8192 &&
           (double)K * (double)F * 4.0 < 2.0e9 && (double)K * (double)K * 4.0 * 64.0 < 2.0e9;
}

// Workgroups the device keeps resident for this kernel (the grid of a kernel with grid barriers must not
// exceed it).  The occupancy query can over-report by one block per CU (guide), hence the margin: two blocks
// per CU are only counted on when the query answers three or more.
inline int fused_update_resident_cap() {
    static int cap[32] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 32) dev = 0;
    if (cap[dev] == 0) {
        hipDeviceProp_t prop;
        int cus = 256, occ = 1;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, nmf_fused_update_kernel<0>, 256, 0) != hipSuccess)
            occ = 1;
        cap[dev] = cus * (occ >= 3 ? 2 : 1);
    }
    return cap[dev];
}

inline FusedUpdPlan fused_update_plan(int64_t F, int64_t K) {
    FusedUpdPlan p;
    p.tiles_k = (int)((K + 63) / 64);
    p.tiles_f = (int)((F + 63) / 64);
    const long ntiles = (long)p.tiles_k * p.tiles_f;
    const int cap = fused_update_resident_cap();
    p.grid = (int)(ntiles < cap ? ntiles : cap);
    if (p.grid < 1) p.grid = 1;
    const long gtiles = (long)p.tiles_k * p.tiles_k;
    long slices = p.grid / gtiles;
    if (slices < 1) slices = 1;
    if (slices > p.tiles_f) slices = p.tiles_f;
    if (slices > 64) slices = 64;
    long cols = ((F + slices - 1) / slices + 63) / 64 * 64;
    p.slice_cols = (int)cols;
    p.gram_slices = (int)((F + cols - 1) / cols);
    p.U_elems = (size_t)K * F;
    p.rowpart_elems = (size_t)K * p.tiles_f;
    p.gslab_elems = (size_t)p.gram_slices * K * K;
    p.wgmax_elems = (size_t)p.grid;
    return p;
}

struct FusedUpdWs {
    float* U = nullptr;
    float* Sred = nullptr;
    float* rowpart = nullptr;
    float* gslabs = nullptr;
    float* wgmax = nullptr;
};
inline void fused_update_ws_plan(WsPlan& plan, int64_t F, int64_t K) {
    const FusedUpdPlan p = fused_update_plan(F, K);
    plan.add<float>(p.U_elems);
    plan.add<float>((size_t)K * K);
    plan.add<float>(p.rowpart_elems);
    plan.add<float>(p.gslab_elems);
    plan.add<float>(p.wgmax_elems);
}
inline int fused_update_ws_carve(dcp_handle* h, FusedUpdWs& w, int64_t F, int64_t K) {
    const FusedUpdPlan p = fused_update_plan(F, K);
    w.U = ws_alloc<float>(h, p.U_elems);
    w.Sred = ws_alloc<float>(h, (size_t)K * K);
    w.rowpart = ws_alloc<float>(h, p.rowpart_elems);
    w.gslabs = ws_alloc<float>(h, p.gslab_elems);
    w.wgmax = ws_alloc<float>(h, p.wgmax_elems);
    if (!w.U || !w.Sred || !w.rowpart || !w.gslabs || !w.wgmax)
        return fail(h, DCP_ERR_INTERNAL, "fused update workspace plan mismatch");
    return DCP_OK;
}

// The handle's barrier state: a small device buffer of its own (never part of the arena, which other calls
// scribble over), zeroed once; the kernel leaves it zeroed.
inline int fused_update_barrier(dcp_handle* h, GridBarrierState** out) {
    if (h->grid_barrier == nullptr) {
        void* p = nullptr;
        DCP_HIP_OK(h, hipMalloc(&p, sizeof(GridBarrierState)));
        hipError_t e = hipMemset(p, 0, sizeof(GridBarrierState));
        if (e != hipSuccess) {
            (void)hipFree(p);
            return fail(h, DCP_ERR_HIP, "grid barrier state memset failed");
        }
        h->grid_barrier = p;
    }
    *out = static_cast<GridBarrierState*>(h->grid_barrier);
    return DCP_OK;
}

// A barrier wait of the fused launch expired (its grid was not fully resident): clear the sticky state so
// that the handle stays usable and report the failure.
inline int fused_update_expired(dcp_handle* h) {
    (void)hipStreamSynchronize(h->stream);
    if (h->grid_barrier != nullptr) (void)hipMemset(h->grid_barrier, 0, sizeof(GridBarrierState));
    return fail(h, DCP_ERR_INTERNAL, "fused D-side launch: a grid-barrier wait expired");
}

// stats: reduced [K, F + K] (nslabs == 0) or split-K partials (nslabs >= 1, slab_stride elements apart).
// Writes D_new, G = D_new D_new^T, *maxdiff_out (device or pinned host), clears *maxdiff_zero when given.
inline int nmf_fused_update(dcp_handle* h, const float* stats, int nslabs, long slab_stride, const float* D,
                            float* D_new, float* G, int64_t F, int64_t K, float* maxdiff_out,
                            float* maxdiff_zero, int* status_out, FusedUpdWs& w) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(stats) || !al16(D) || !al16(D_new) || !al16(G) || (nslabs >= 1 && slab_stride % 4 != 0))
        return fail(h, DCP_ERR_INTERNAL, "fused update: unaligned operand");
    const FusedUpdPlan p = fused_update_plan(F, K);
    FusedUpdArgs a;
    a.stats = stats; a.slab_stride = slab_stride; a.nslabs = nslabs; a.W = (int)(F + K);
    a.D = D; a.D_new = D_new; a.U = w.U; a.Sred = w.Sred; a.rowpart = w.rowpart; a.gslabs = w.gslabs;
    a.G = G; a.wgmax = w.wgmax; a.maxdiff_out = maxdiff_out; a.maxdiff_zero = maxdiff_zero;
    a.status_out = status_out;
    DCP_TRY(fused_update_barrier(h, &a.bar));
    a.K = (int)K; a.F = (int)F;
    a.tiles_k = p.tiles_k; a.tiles_f = p.tiles_f; a.gram_slices = p.gram_slices; a.slice_cols = p.slice_cols;
    a.stamps = nullptr;
    static const bool want_stamps = getenv("DCP_FUSED_STAMPS") != nullptr;
    static unsigned long long* stamp_buf = nullptr;
    if (want_stamps) {
        if (stamp_buf == nullptr) (void)hipMalloc(reinterpret_cast<void**>(&stamp_buf), 32 * sizeof(unsigned long long));
        a.stamps = stamp_buf;
    }
    hipLaunchKernelGGL((nmf_fused_update_kernel<0>), dim3(p.grid), dim3(256), 0, h->stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, DCP_ERR_HIP, std::string("launch failed: ") + hipGetErrorString(e));
    if (want_stamps && stamp_buf != nullptr) {   // analysis only: synchronous
        unsigned long long hs[32];
        (void)hipStreamSynchronize(h->stream);
        (void)hipMemcpy(hs, stamp_buf, sizeof(hs), hipMemcpyDeviceToHost);
        static int printed = 0;
        if (printed++ % 16 == 8) {
            fprintf(stderr, "[fused stamps, us since kernel start of block 0] grid=%d K=%d F=%d nslabs=%d\n", p.grid, (int)K, (int)F, nslabs);
            for (int b = 0; b < 2; ++b) {
                fprintf(stderr, "  block %s:", b == 0 ? "0   " : "last");
                for (int i = 0; i <= 10; ++i) fprintf(stderr, " %7.2f", (double)(long long)(hs[b * 16 + i] - hs[0]) * 0.01);
                fprintf(stderr, "\n");
            }
        }
    }
    return DCP_OK;
}

}  // namespace dcp
