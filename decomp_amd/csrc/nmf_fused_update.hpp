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
};

constexpr int kFuLd = 68;   // LDS row stride in floats (16-byte aligned rows)

// acc += A[arow0 + r, k] * B(k, n) for the 32 x 32 sub-tile of wave (wm, wn) of a 64 x 64 tile, k in [k0, k1).
// A: row-major [rows, lda] (reduction index contiguous).  B: BK == KMAJOR: row-major [n, ldb] with the
// reduction index contiguous (A . B^T); BK == XMAJOR: row-major [k, ldb] (A . B).  Rows / columns past the
// valid counts and k >= k1 contribute zeros.  All leading dims, k0, k1 and column origins are multiples of 4
// and the bases 16-byte aligned (checked by the launcher).
template <int BLAY>
__device__ __forceinline__ void fu_mma64(f32x16& acc, const float* __restrict__ A, long lda, int arow0,
                                         int arows, const float* __restrict__ B, long ldb, int b0, int bvalid,
                                         int k0, int k1, float* sA, float* sB, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    f32x4 ra[4], rb[4];
    auto gload = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx >> 4, kq = (idx & 15) << 2;
            {   // A chunk: 64 rows x 64 k
                const int row = arow0 + r;
                const bool ok = (row < arows) && (kc + kq < k1);
                const float* p = A + (long)(ok ? row : 0) * lda + (ok ? kc + kq : 0);
                const f32x4 v = *reinterpret_cast<const f32x4*>(p);
                ra[i] = ok ? v : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
            if (BLAY == KMAJOR) {   // B chunk: 64 n-rows x 64 k
                const int row = b0 + r;
                const bool ok = (row < bvalid) && (kc + kq < k1);
                const float* p = B + (long)(ok ? row : 0) * ldb + (ok ? kc + kq : 0);
                const f32x4 v = *reinterpret_cast<const f32x4*>(p);
                rb[i] = ok ? v : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            } else {                // B chunk: 64 k-rows x 64 n
                const int kr = kc + r, col = b0 + kq;
                const bool ok = (kr < k1) && (col < bvalid);
                const float* p = B + (long)(ok ? kr : 0) * ldb + (ok ? col : 0);
                const f32x4 v = *reinterpret_cast<const f32x4*>(p);
                rb[i] = ok ? v : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
    };
    if (k0 >= k1) return;
    gload(k0);
    for (int kc = k0; kc < k1; kc += 64) {
        __syncthreads();   // the previous chunk's fragment reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx >> 4, kq = (idx & 15) << 2;
            *reinterpret_cast<f32x4*>(sA + r * kFuLd + kq) = ra[i];
            *reinterpret_cast<f32x4*>(sB + r * kFuLd + kq) = rb[i];
        }
        __syncthreads();
        if (kc + 64 < k1) gload(kc + 64);   // lands under this chunk's MFMAs
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(sA + (wm * 32 + l31) * kFuLd + 8 * c + 4 * h);
            f32x4 fb;
            if (BLAY == KMAJOR) {
                fb = *reinterpret_cast<const f32x4*>(sB + (wn * 32 + l31) * kFuLd + 8 * c + 4 * h);
            } else {
                const float* q = sB + (8 * c + 4 * h) * kFuLd + wn * 32 + l31;
                fb[0] = q[0];
                fb[1] = q[kFuLd];
                fb[2] = q[2 * kFuLd];
                fb[3] = q[3 * kFuLd];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc, 0, 0, 0);
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

// ordered sum over the statistics slabs of 4 consecutive elements starting at offset `o` of a slab
__device__ __forceinline__ f32x4 fu_slab_sum4(const float* __restrict__ base, long o, long stride, int S) {
    f32x4 acc = *reinterpret_cast<const f32x4*>(base + o);
    int s = 1;
    for (; s + 7 < S; s += 8) {   // same left-to-right order as reduce_slabs_kernel, eight loads in flight
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + (long)(s + u) * stride + o);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = acc[e] + v[u][e];
    }
    for (; s < S; ++s) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long)s * stride + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] + v[e];
    }
    return acc;
}

// (a template only so that the header may be included by several translation units)
template <int VARIANT>
__global__ void __launch_bounds__(256, 1) nmf_fused_update_kernel(FusedUpdArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[64 * kFuLd];
    __shared__ __attribute__((aligned(16))) float sB[64 * kFuLd];
    __shared__ __attribute__((aligned(16))) float sU[64 * kFuLd];
    __shared__ float s_red[4];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const int K = a.K, F = a.F, W = a.W;
    const long KK = (long)K * K;
    const int ntiles = a.tiles_k * a.tiles_f;
    unsigned epoch = 0;

    // ---- A: S = sum of the slabs' x^T x block ------------------------------------------------
    const float* S = a.stats + F;   // reduced statistics: S at column F, leading dim W
    long ldS = W;
    if (a.nslabs >= 1) {
        for (long i4 = (long)blockIdx.x * 256 + tid; i4 < KK / 4; i4 += (long)gridDim.x * 256) {
            const long e = i4 * 4;
            const long row = e / K, col = e - row * K;
            const f32x4 v = fu_slab_sum4(a.stats, row * W + F + col, a.slab_stride, a.nslabs);
            *reinterpret_cast<f32x4*>(a.Sred + e) = v;
        }
        S = a.Sred;
        ldS = K;
        grid_barrier(a.bar, ++epoch);
    }

    // ---- B: U tile = D o max(num, 0) / max(S D, eps), row partial sums of U^2 ---------------------
    const int row_t = tid >> 2, q = tid & 3;   // epilogue thread map: row of the tile, 16-column quarter
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int ti = tile % a.tiles_k, tf = tile / a.tiles_k;
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
            const long o = (long)(ok ? grow : 0) * W + (ok ? gcol : 0);
            const f32x4 nu = (a.nslabs >= 1) ? fu_slab_sum4(a.stats, o, a.slab_stride, a.nslabs)
                                             : *reinterpret_cast<const f32x4*>(a.stats + o);
            const f32x4 d = *reinterpret_cast<const f32x4*>(a.D + (long)(ok ? grow : 0) * F + (ok ? gcol : 0));
            const f32x4 den = *reinterpret_cast<const f32x4*>(sU + row_t * kFuLd + q * 16 + 4 * i);
            f32x4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u[e] = d[e] * max_np(nu[e], 0.0f) / max_np(den[e], 1.0e-15f);   // grads.py:93
                if (!ok) u[e] = 0.0f;
                ss += u[e] * u[e];
            }
            if (ok) *reinterpret_cast<f32x4*>(a.U + (long)grow * F + gcol) = u;
        }
        ss = ss + __shfl_xor(ss, 1, 64);   // (q0 + q1), (q2 + q3)
        ss = ss + __shfl_xor(ss, 2, 64);   // ((q0 + q1) + (q2 + q3)) on every lane of the quad
        if (q == 0 && grow < K) a.rowpart[(long)grow * a.tiles_f + tf] = ss;
        __syncthreads();   // sU is rewritten by the next tile
    }
    grid_barrier(a.bar, ++epoch);

    // ---- C: D_new = U / |U|, max|D - D_new| ---------------------------------------------------------------
    float md = 0.0f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int ti = tile % a.tiles_k, tf = tile / a.tiles_k;
        const int grow = ti * 64 + row_t;
        const bool rok = grow < K;
        float tot = 0.0f;
        for (int j = q; j < a.tiles_f; j += 4) tot += a.rowpart[(long)(rok ? grow : 0) * a.tiles_f + j];
        tot = tot + __shfl_xor(tot, 1, 64);
        tot = tot + __shfl_xor(tot, 2, 64);
        const float nrm = sqrtf(tot);           // normalize.py:13-21 (l2_strict)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gcol = tf * 64 + q * 16 + 4 * i;
            if (rok && gcol < F) {
                const long o = (long)grow * F + gcol;
                const f32x4 u = *reinterpret_cast<const f32x4*>(a.U + o);
                const f32x4 d = *reinterpret_cast<const f32x4*>(a.D + o);
                f32x4 dn;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dn[e] = u[e] / nrm;                         // a true division, as the reference's U / sqrt(.)
                    const float df = fabsf(d[e] - dn[e]);
                    md = (df > md || df != df) ? df : md;       // NaN propagates, as np.max
                }
                *reinterpret_cast<f32x4*>(a.D_new + o) = dn;
            }
        }
    }
    {
        const float m = block_max_256(md, s_red);
        if (tid == 0) a.wgmax[blockIdx.x] = m;
    }
    grid_barrier(a.bar, ++epoch);

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
                *reinterpret_cast<f32x4*>(a.gslabs + (long)slice * KK + (long)grow * K + gcol) =
                    *reinterpret_cast<const f32x4*>(sU + row_t * kFuLd + q * 16 + 4 * i);
        }
        __syncthreads();
    }
    grid_barrier(a.bar, ++epoch);

    // ---- E: G = ordered sum of the partials; max|D - D_new| over the workgroups ---------------------------
    for (long i4 = (long)blockIdx.x * 256 + tid; i4 < KK / 4; i4 += (long)gridDim.x * 256)
        *reinterpret_cast<f32x4*>(a.G + i4 * 4) = fu_slab_sum4(a.gslabs, i4 * 4, KK, a.gram_slices);
    if (blockIdx.x == 0) {
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
    (void)grid_barrier_finish(a.bar, &s_last);
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
    return !off && (K % 4 == 0) && (F % 4 == 0) && K >= 4 && F >= 4 && K <= 8192;
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
    hipLaunchKernelGGL((nmf_fused_update_kernel<0>), dim3(p.grid), dim3(256), 0, h->stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, DCP_ERR_HIP, std::string("launch failed: ") + hipGetErrorString(e));
    return DCP_OK;
}

}  // namespace dcp
