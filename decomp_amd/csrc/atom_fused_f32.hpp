// float32 fast path of the blocked atom sweep (atom_sweep.hpp) for K and F multiples of 64: three launches
// per 64-atom block instead of seven, and no cross-stream events.
//
//   K1  atom_recur_lookahead_kernel   grid = 1 + F/64 workgroups of 256 threads, ONE launch:
//         workgroup 0        the block's 64 dependent steps (atom_recur_body)
//         workgroups 1..     the K-deep part of the NEXT block's P (everything but this block's atoms, which
//                            do not exist yet): a 64 x 64 output tile each, 64-deep K units, MFMA
//       (the recursion occupies one CU; the product fills others meanwhile -- the same overlap as two
//        streams, without the ~6 us an event costs on each side of it)
//   K2  atom_apply_kernel             grid = F/64: per 64-column tile, three chained 64x64x64 products
//         T = E . P[:, tile]  -> D_new[block][:, tile]        (dictionary_learning.py:159 in coefficient space)
//         Pn = Ppart + Aprev . T -> P_next[:, tile]           (this block's contribution to the next block)
//         Gs = Pn . Pn^T      -> slab[tile]                   (the next block's Gram matrix, one slab per tile)
//       operands stay in LDS between the products
//   K3  reduce_slabs_kernel           G_next = sum of the F/64 slabs (ordered)
// Same arithmetic as the generic path (products split at the same places); k-order inside a product is
// the MFMA core's fixed permutation.
#pragma once
#include "atom_sweep.hpp"

namespace dcp {

struct AtomFusedArgs {
    // recursion of the current block
    const float* G;       // [64, 64]
    const float* Wl;      // [64, 64]  (rows of this block)
    float* E;             // [64, 64]
    // look-ahead product of the next block (has_next)
    const float* Alook;   // rows of the next block, [64, K]
    const float* Dcur;    // [K, F]  D_new as it stands
    const float* Bn;      // B rows of the next block [64, F]
    const float* Dold;    // D_new rows of the next block (not yet updated) [64, F]
    const float* rden;    // [64] of the next block
    float* Pnext;         // [64, F]
    int K, F;
    int has_next;
};

// 64 x 64 tile, 4 waves of 32 x 32, operands from LDS images of PanelGeom<., 64, 64, 256>.
template <int ALAY, int BLAY>
__device__ __forceinline__ void mma_64x64x64(f32x16& acc, const float* sA, const float* sB, int wm, int wn,
                                             int l31, int h) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 fa = panel_frag<ALAY, 64, 64>(sA, wm * 32 + l31, c, h);
        const f32x4 fb = panel_frag<BLAY, 64, 64>(sB, wn * 32 + l31, c, h);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc, 0, 0, 0);
    }
}

// (templates only so that the header can be included by every dtype's translation unit)
// 512 threads: the recursion's trailing updates are shared by 8 waves (4: 7.8 us of a 20 us block, in-kernel stamps);
// the look-ahead workgroups run on their first 256 threads (the others leave before any barrier).
constexpr int kAtomRecurThreads = 512;
template <class T = float>
__global__ void __launch_bounds__(kAtomRecurThreads) atom_recur_lookahead_kernel(AtomFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
    if (blockIdx.x == 0) {
        atom_recur_body<float, kAtomRecurThreads>(fused_lds, 64, a.G, a.Wl, a.E);
        return;
    }
    if (!a.has_next || threadIdx.x >= 256) return;
    // ---- P_next[:, tile] = (Bn - Alook . Dcur) * rden + Dold      (EpiAtomP of the generic path) ----
    typedef PanelGeom<KMAJOR, 64, 64, 256> GA;   // Alook rows: [64][64 k]
    typedef PanelGeom<XMAJOR, 64, 64, 256> GB;   // Dcur rows:  [64 k][64 cols]
    float* smem = reinterpret_cast<float*>(fused_lds);
    float* sA0 = smem;
    float* sB0 = smem + 2 * GA::ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int n0 = (blockIdx.x - 1) * 64;
    const int nkb = a.K / 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    f32x4 ra[GA::F4], rb[GB::F4];
    panel_gload<KMAJOR, 64, 64, false, 256>(ra, a.Alook, (long)a.K, 0, 64, 0, a.K, tid);
    panel_gload<XMAJOR, 64, 64, false, 256>(rb, a.Dcur, (long)a.F, n0, a.F, 0, a.K, tid);
    panel_lds_store<KMAJOR, 64, 64, 256>(sA0, ra, tid);
    panel_lds_store<XMAJOR, 64, 64, 256>(sB0, rb, tid);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        const bool more = (kb + 1) < nkb;
        if (more) {
            panel_gload<KMAJOR, 64, 64, false, 256>(ra, a.Alook, (long)a.K, 0, 64, (kb + 1) * 64, a.K, tid);
            panel_gload<XMAJOR, 64, 64, false, 256>(rb, a.Dcur, (long)a.F, n0, a.F, (kb + 1) * 64, a.K, tid);
        }
        mma_64x64x64<KMAJOR, XMAJOR>(acc, sA0 + cur * GA::ELEMS, sB0 + cur * GB::ELEMS, wm, wn, l31, h);
        if (more) {
            panel_lds_store<KMAJOR, 64, 64, 256>(sA0 + (cur ^ 1) * GA::ELEMS, ra, tid);
            panel_lds_store<XMAJOR, 64, 64, 256>(sB0 + (cur ^ 1) * GB::ELEMS, rb, tid);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const long i = (long)row * a.F + col;
        a.Pnext[i] = (a.Bn[i] - acc[r]) * a.rden[row] + a.Dold[i];
    }
}

struct AtomApplyArgs {
    const float* E;       // [64, 64]
    const float* P;       // [64, F]   this block's primed vectors
    float* Dblk;          // D_new + k0 * F : rows of this block (output)
    const float* Aprev;   // [64, 64]  -rden * A[next block, this block]   (has_next)
    float* Pnext;         // [64, F]   in: look-ahead part, out: complete P of the next block
    float* slabs;         // [F / 64][64 * 64]  Gram slabs of the next block
    int F;
    int has_next;
};

template <class T = float>
__global__ void __launch_bounds__(256) atom_apply_kernel(AtomApplyArgs a) {
    typedef PanelGeom<KMAJOR, 64, 64, 256> GK;
    __shared__ __attribute__((aligned(16))) float sE[64 * 64];    // K-major image of E (A operand)
    __shared__ __attribute__((aligned(16))) float sP[64 * 64];    // [k = row of P][col]  (B operand, X-major)
    __shared__ __attribute__((aligned(16))) float sT[64 * 64];    // first Aprev (K-major), later Pn (K-major)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 64;
    f32x4 r4[GK::F4];
    panel_gload<KMAJOR, 64, 64, false, 256>(r4, a.E, 64L, 0, 64, 0, 64, tid);
    panel_lds_store<KMAJOR, 64, 64, 256>(sE, r4, tid);
    panel_gload<XMAJOR, 64, 64, false, 256>(r4, a.P, (long)a.F, n0, a.F, 0, 64, tid);
    panel_lds_store<XMAJOR, 64, 64, 256>(sP, r4, tid);
    if (a.has_next) {
        panel_gload<KMAJOR, 64, 64, false, 256>(r4, a.Aprev, 64L, 0, 64, 0, 64, tid);
        panel_lds_store<KMAJOR, 64, 64, 256>(sT, r4, tid);
    }
    __syncthreads();
    // ---- T = E . P[:, tile] ----
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    mma_64x64x64<KMAJOR, XMAJOR>(acc, sE, sP, wm, wn, l31, h);
    const int cl = wn * 32 + l31;                 // tile-local column
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        a.Dblk[(long)row * a.F + n0 + cl] = acc[r];
    }
    if (!a.has_next) return;
    __syncthreads();                              // every wave is done reading sP
#pragma unroll
    for (int r = 0; r < 16; ++r) {                // T as the next product's B operand: [k = row][col]
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        sP[row * 64 + cl] = acc[r];
    }
    __syncthreads();
    // ---- Pn = Ppart + Aprev . T ----
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    mma_64x64x64<KMAJOR, XMAJOR>(acc, sT, sP, wm, wn, l31, h);
    __syncthreads();                              // sT (Aprev) is dead: it becomes the K-major image of Pn
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const long i = (long)row * a.F + n0 + cl;
        const float pn = a.Pnext[i] + acc[r];
        a.Pnext[i] = pn;
        sT[GK::kchunk(row, cl >> 2) + (cl & 3)] = pn;     // element (row, k = column)
    }
    __syncthreads();
    // ---- Gs = Pn . Pn^T over this tile's 64 columns ----
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    mma_64x64x64<KMAJOR, KMAJOR>(acc, sT, sT, wm, wn, l31, h);
    float* slab = a.slabs + (long)blockIdx.x * 64 * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        slab[row * 64 + cl] = acc[r];
    }
}

// G = sum of S slabs of 64 x 64 floats, two levels in a FIXED order (bitwise reproducible): workgroup b owns elements
// 64 b .. 64 b + 63; wave q sums slabs q, q + 4, q + 8, ... (in that order, sixteen loads in flight) into LDS, then the
// first wave adds the four partial sums in the order q = 0, 1, 2, 3.  4 x the parallelism of the flat ordered sum on a
// latency-bound 1 MB read (5.3 -> ~3 us per block of the atom sweep).
template <class T = float>
__global__ void __launch_bounds__(256) gram_slab_sum_kernel(const float* __restrict__ slabs, int S, long stride,
                                                            float* __restrict__ G) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long i = blockIdx.x * 64L + lane;
    float acc = 0.0f;
    int s = q;
    for (; s + 60 < S; s += 64) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = slabs[(long)(s + 4 * u) * stride + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    for (; s < S; s += 4) acc += slabs[(long)s * stride + i];
    part[q][lane] = acc;
    __syncthreads();
    if (q == 0) G[i] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

inline bool atom_fused_ok(int64_t F, int64_t K) { return K >= 64 && (K % 64) == 0 && F >= 64 && (F % 64) == 0; }

// D_new (a copy of D on entry) <- the swept dictionary: the float32 fast path.  Same workspace as atom_sweep.
inline int atom_sweep_fused_f32(dcp_handle* h, const float* A, const float* B, float* Dnew, int64_t F64,
                                int64_t K64, AtomWs<float>& w) {
    hipStream_t st = h->stream;
    const int K = (int)K64, F = (int)F64;
    const int nblk = K / 64, ntile = F / 64;
    if ((size_t)ntile * 64 * 64 > w.slab_count) return fail(h, DCP_ERR_INTERNAL, "atom slab plan");
    hipLaunchKernelGGL((atom_prep_kernel<float>), dim3(grid_for((long)K * K, 256)), dim3(256), 0, st, 64, K, A,
                       w.Ablk, w.Wl, w.rden, w.Alook, w.Aprev, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    DCP_LAUNCH_OK(h, hipGetLastError());
    {
        static DynLdsRaised raised;
        std::atomic<bool>& r = raised.on_current_device();
        if (!r) {
            DCP_LAUNCH_OK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&atom_recur_lookahead_kernel<float>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)atom_recur_lds_bytes<float>()));
            r = true;
        }
    }
    {   // block 0: P and its Gram matrix by the generic products
        GemmArgs<float> a;
        a.A = w.Ablk; a.lda = K; a.B = Dnew; a.ldb = F; a.M = 64; a.N = F; a.K = K;
        a.tile = TILE_SMALL_DEEP;
        DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a, EpiAtomP<float>{B, Dnew, w.rden, w.P, (long)F})));
        GemmArgs<float> g;
        g.A = w.P; g.lda = F; g.B = w.P; g.ldb = F; g.M = 64; g.N = 64; g.K = F;
        g.tile = TILE_SMALL_DEEP;
        plan_splits<FORM_NT>(g, 64, 64, 4);
        if ((size_t)g.ksplits * 64 * 64 > w.slab_count) return fail(h, DCP_ERR_INTERNAL, "atom slab plan");
        DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, g, EpiSlab<float>{w.slabs, 64L, 64L * 64})));
        hipLaunchKernelGGL((reduce_slabs_kernel<float>), dim3(16), dim3(256), 0, st, (const float*)w.slabs,
                           64L * 64, g.ksplits, 64L * 64, w.G);
        DCP_LAUNCH_OK(h, hipGetLastError());
    }
    for (int b = 0; b < nblk; ++b) {
        const int k0 = b * 64, k1 = k0 + 64;
        const bool has_next = (b + 1) < nblk;
        float* P = w.P + (size_t)(b & 1) * kAtomBlkMax * F;
        float* Pnext = w.P + (size_t)((b + 1) & 1) * kAtomBlkMax * F;
        AtomFusedArgs fa;
        fa.G = w.G; fa.Wl = w.Wl + (long)k0 * kAtomBlkMax; fa.E = w.E;
        fa.Alook = has_next ? w.Alook + (long)k1 * K : nullptr;
        fa.Dcur = Dnew;
        fa.Bn = has_next ? B + (long)k1 * F : nullptr;
        fa.Dold = has_next ? Dnew + (long)k1 * F : nullptr;
        fa.rden = has_next ? w.rden + k1 : nullptr;
        fa.Pnext = Pnext; fa.K = K; fa.F = F; fa.has_next = has_next ? 1 : 0;
        hipLaunchKernelGGL((atom_recur_lookahead_kernel<float>), dim3(has_next ? 1 + ntile : 1),
                           dim3(kAtomRecurThreads), atom_recur_lds_bytes<float>(), st, fa);
        DCP_LAUNCH_OK(h, hipGetLastError());
        AtomApplyArgs aa;
        aa.E = w.E; aa.P = P; aa.Dblk = Dnew + (long)k0 * F;
        aa.Aprev = has_next ? w.Aprev + (long)k1 * kAtomBlkMax : nullptr;
        aa.Pnext = Pnext; aa.slabs = w.slabs; aa.F = F; aa.has_next = has_next ? 1 : 0;
        hipLaunchKernelGGL((atom_apply_kernel<float>), dim3(ntile), dim3(256), 0, st, aa);
        DCP_LAUNCH_OK(h, hipGetLastError());
        if (has_next) {
            hipLaunchKernelGGL((gram_slab_sum_kernel<float>), dim3(64), dim3(256), 0, st, (const float*)w.slabs, ntile,
                               64L * 64, w.G);
            DCP_LAUNCH_OK(h, hipGetLastError());
        }
    }
    return DCP_OK;
}

}  // namespace dcp
