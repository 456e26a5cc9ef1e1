// complex64 fast path of the blocked atom sweep (atom_sweep.hpp) for K a multiple of 32 and F a multiple of 64:
// the float32 fused structure (atom_fused_f32.hpp: three launches per block, no cross-stream events, operands
// staying in LDS between the chained products) on REAL images of the complex operands.
//
// A block of 32 complex atoms is 64 real rows: with the rows of every F-long operand held PLANAR
// (row 2k = Re, row 2k + 1 = Im of complex row k) a complex left-multiplication  C = A . B  is the real product
//     C^ = A~ . B^ ,   A~[2m, 2j] = Re a_mj,  A~[2m, 2j+1] = -Im a_mj,  A~[2m+1, 2j] = Im a_mj,  A~[2m+1, 2j+1] = Re a_mj
// so the three chained products of a block (T = E.P, Pn = Ppart + Aprev.T, Gs = Pn.Pn^H) and the K-deep look-ahead
// product are exactly the float32 kernels' 64 x 64 x 64 MFMA tiles -- no per-launch extended / planar images of the
// F-long operands, which is where the generic complex path (16 blocks x 5 launches + 19 image kernels at K = 512)
// spends most of its time.  The Hermitian Gram matrix comes out of the real one,
//     (P P^H)[a, b] = (G^[2a, 2b] + G^[2a+1, 2b+1]) + i (G^[2a+1, 2b] - G^[2a, 2b+1]),
// inside the ordered slab sum; the 32 dependent steps per block stay atom_recur_body<c64> (complex double).
// Reference: decomp/dictionary_learning.py:154-159; same re-association as atom_sweep.hpp, rounding level identical
// (fp32 MFMA products, coefficient recursion in double).
#pragma once
#include "atom_fused_f32.hpp"

namespace dcp {

// Dhat[2k, f] = Re D[k, f], Dhat[2k + 1, f] = Im D[k, f]
template <class T = c64>
__global__ void __launch_bounds__(256) c64_planar_rows_kernel(const c64* __restrict__ D, long K, long F,
                                                              float* __restrict__ Dhat) {
    const long n = K * F;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const long k = i / F, f = i - k * F;
        const c64 v = D[i];
        Dhat[(2 * k) * F + f] = v.re;
        Dhat[(2 * k + 1) * F + f] = v.im;
    }
}

// out [2 rows, 2 cols] (leading dim ldo) = the real left-multiplication image A~ of A [rows, cols] (leading dim lda)
template <class T = c64>
__global__ void __launch_bounds__(256) c64_extend_left_kernel(const c64* __restrict__ A, long rows, long cols, long lda,
                                                              float* __restrict__ out, long ldo) {
    const long n = rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const long r = i / cols, j = i - r * cols;
        const c64 v = A[r * lda + j];
        float* o = out + (2 * r) * ldo + 2 * j;
        o[0] = v.re;
        o[1] = -v.im;
        o[ldo] = v.im;
        o[ldo + 1] = v.re;
    }
}

// G [32, 32] complex = the Hermitian combination of the sum of S real 64 x 64 slabs (stride apart), summed in a FIXED
// two-level order (bitwise reproducible).  32 workgroups x 512 threads: workgroup a owns complex row a = real rows
// 2a, 2a + 1 (128 contiguous floats of a slab); thread group q = tid / 128 sums slabs q, q + 4, ... in that order
// (sixteen loads in flight), the four partial sums are added in the order q = 0..3, and 32 threads combine.  4 x the
// parallelism of c64_gram_combine_kernel's flat ordered sum on a latency-bound 2 MB read.
template <class T = c64>
__global__ void __launch_bounds__(512) c64_gram_combine_tree_kernel(const float* __restrict__ slabs, int S, long stride,
                                                                    c64* __restrict__ G) {
    __shared__ float part[4][128];
    __shared__ float row[128];
    const int tid = threadIdx.x, e = tid & 127, q = tid >> 7;
    const long i = blockIdx.x * 128L + e;
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
    part[q][e] = acc;
    __syncthreads();
    if (tid < 128) row[tid] = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
    __syncthreads();
    if (tid < 32) {
        const int bb = tid;
        const float s00 = row[2 * bb], s01 = row[2 * bb + 1], s10 = row[64 + 2 * bb], s11 = row[64 + 2 * bb + 1];
        G[blockIdx.x * 32 + bb] = c64{s00 + s11, s10 - s01};
    }
}

// G [32, 32] complex = the Hermitian combination of the ordered sum of S real 64 x 64 slabs (stride apart).
// 16 workgroups x 256 threads: a thread sums ONE real element over the slabs in their fixed order (sixteen loads in
// flight; one at a time the loop is a chain of L2 round trips), a workgroup holds four consecutive real rows =
// two complex rows, combined through LDS.
template <class T = c64>
__global__ void __launch_bounds__(256) c64_gram_combine_kernel(const float* __restrict__ slabs, int S, long stride,
                                                               c64* __restrict__ G) {
    __shared__ float sh[256];
    const int tid = threadIdx.x;
    const long i = blockIdx.x * 256L + tid;            // element of the 64 x 64 real Gram matrix
    float acc = slabs[i];
    int s = 1;
    for (; s + 15 < S; s += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = slabs[(long)(s + u) * stride + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    for (; s < S; ++s) acc += slabs[(long)s * stride + i];
    sh[tid] = acc;
    __syncthreads();
    if (tid < 64) {                                     // rows 4 q .. 4 q + 3 of this workgroup: a = 2 q + (tid >> 5)
        const int al = tid >> 5, b = tid & 31;
        const float s00 = sh[(2 * al) * 64 + 2 * b], s01 = sh[(2 * al) * 64 + 2 * b + 1];
        const float s10 = sh[(2 * al + 1) * 64 + 2 * b], s11 = sh[(2 * al + 1) * 64 + 2 * b + 1];
        G[(2 * blockIdx.x + al) * 32 + b] = c64{s00 + s11, s10 - s01};
    }
}

struct AtomFusedArgsC {
    // recursion of the current block (G == nullptr: no recursion, look-ahead product only)
    const c64* G;         // [32, 32]
    const c64* Wl;        // [32, kAtomBlkMax]  (rows of this block)
    c64* E;               // [32, 32]
    float* E_ext;         // [64, 64]  the real left-multiplication image of E
    // look-ahead product of the next block (has_next)
    const float* Alook;   // extended rows of the next block, [64, 2K]
    const float* Dhat;    // [2K, F]  planar D_new as it stands
    const c64* Bn;        // B rows of the next block [32, F]
    const c64* Dold;      // D_new rows of the next block (not yet updated) [32, F]
    const c64* rden;      // [32] of the next block
    float* Pnext;         // planar [64, F]
    float* Pnext2;        // ksplit == 2: the second half of the reduction lands here as -acc * rden (summed by the apply kernel)
    int K2, F;            // K2 = 2 K
    int has_next;
    int ksplit;           // 1 or 2 workgroups per column tile (the 2K-deep product on 2 F/64 CUs instead of F/64)
};

template <class T = c64>
__global__ void __launch_bounds__(kAtomRecurThreads) atom_recur_lookahead_c64_kernel(AtomFusedArgsC a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
    if (blockIdx.x == 0) {
        if (a.G != nullptr)
            atom_recur_body<c64, kAtomRecurThreads>(fused_lds, 32, a.G, a.Wl, a.E, (float*)nullptr, a.E_ext);
        return;
    }
    if (!a.has_next || threadIdx.x >= 256) return;
    // ---- planar P_next[:, tile] = (Bn - Alook . Dcur) * rden + Dold      (EpiAtomP of the generic path) ----
    typedef PanelGeom<KMAJOR, 64, 64, 256> GA;   // extended Alook rows: [64][64 k]
    typedef PanelGeom<XMAJOR, 64, 64, 256> GB;   // planar Dcur rows:    [64 k][64 cols]
    float* smem = reinterpret_cast<float*>(fused_lds);
    float* sA0 = smem;
    float* sB0 = smem + 2 * GA::ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int ntile = a.F / 64;
    const int wg = blockIdx.x - 1;
    const int kh = wg / ntile;                       // which half of the reduction (ksplit == 2)
    const int n0 = (wg - kh * ntile) * 64;
    const int nkb_all = a.K2 / 64;
    const int kb0 = kh * (nkb_all / a.ksplit);
    const int nkb = (a.ksplit == 2) ? nkb_all / 2 : nkb_all;
    const int k00 = kb0 * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    f32x4 ra[GA::F4], rb[GB::F4];
    panel_gload<KMAJOR, 64, 64, false, 256>(ra, a.Alook, (long)a.K2, 0, 64, k00, a.K2, tid);
    panel_gload<XMAJOR, 64, 64, false, 256>(rb, a.Dhat, (long)a.F, n0, a.F, k00, a.K2, tid);
    panel_lds_store<KMAJOR, 64, 64, 256>(sA0, ra, tid);
    panel_lds_store<XMAJOR, 64, 64, 256>(sB0, rb, tid);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        const bool more = (kb + 1) < nkb;
        if (more) {
            panel_gload<KMAJOR, 64, 64, false, 256>(ra, a.Alook, (long)a.K2, 0, 64, k00 + (kb + 1) * 64, a.K2, tid);
            panel_gload<XMAJOR, 64, 64, false, 256>(rb, a.Dhat, (long)a.F, n0, a.F, k00 + (kb + 1) * 64, a.K2, tid);
        }
        mma_64x64x64<KMAJOR, XMAJOR>(acc, sA0 + cur * GA::ELEMS, sB0 + cur * GB::ELEMS, wm, wn, l31, h);
        if (more) {
            panel_lds_store<KMAJOR, 64, 64, 256>(sA0 + (cur ^ 1) * GA::ELEMS, ra, tid);
            panel_lds_store<XMAJOR, 64, 64, 256>(sB0 + (cur ^ 1) * GB::ELEMS, rb, tid);
        }
        __syncthreads();
    }
    // registers r, r + 1 (r even) of a lane are planar rows 2m, 2m + 1: one complex element per pair
    const int col = n0 + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;     // even
        const int m = row >> 1;
        const long i = (long)m * a.F + col;
        const c64 v{acc[r], acc[r + 1]};
        if (kh == 0) {
            const c64 p = add(mul(sub(a.Bn[i], v), a.rden[m]), a.Dold[i]);
            a.Pnext[(long)row * a.F + col] = p.re;
            a.Pnext[(long)(row + 1) * a.F + col] = p.im;
        } else {        // the other half of the sum: (B - v1 - v2) rden + Dold = [(B - v1) rden + Dold] + (-v2 rden)
            const c64 p = mul(c64{-v.re, -v.im}, a.rden[m]);
            a.Pnext2[(long)row * a.F + col] = p.re;
            a.Pnext2[(long)(row + 1) * a.F + col] = p.im;
        }
    }
}

struct AtomApplyArgsC {
    const float* E_ext;   // [64, 64]
    const float* P;       // planar [64, F]   this block's primed vectors
    c64* Dblk;            // D_new + k0 * F : rows of this block (output, complex)
    float* Dhat_blk;      // planar rows of this block in the [2K, F] image (output)
    const float* Aprev;   // [64, 64]  extended -rden * A[next block, this block]   (has_next)
    float* Pnext;         // planar [64, F]   in: look-ahead part, out: complete P of the next block
    const float* Pnext2;  // nullable: the second half of a split look-ahead product (added in)
    float* slabs;         // [F / 64][64 * 64]  real Gram slabs of the next block
    int F;
    int has_next;
};

template <class T = c64>
__global__ void __launch_bounds__(256) atom_apply_c64_kernel(AtomApplyArgsC a) {
    typedef PanelGeom<KMAJOR, 64, 64, 256> GK;
    __shared__ __attribute__((aligned(16))) float sE[64 * 64];    // K-major image of E~ (A operand)
    __shared__ __attribute__((aligned(16))) float sP[64 * 64];    // [k = planar row of P][col]  (B operand, X-major)
    __shared__ __attribute__((aligned(16))) float sT[64 * 64];    // first Aprev~ (K-major), later Pn (K-major)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 64;
    f32x4 r4[GK::F4];
    panel_gload<KMAJOR, 64, 64, false, 256>(r4, a.E_ext, 64L, 0, 64, 0, 64, tid);
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
    for (int r = 0; r < 16; r += 2) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;     // even planar row
        a.Dblk[(long)(row >> 1) * a.F + n0 + cl] = c64{acc[r], acc[r + 1]};
        a.Dhat_blk[(long)row * a.F + n0 + cl] = acc[r];
        a.Dhat_blk[(long)(row + 1) * a.F + n0 + cl] = acc[r + 1];
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
        float pn = a.Pnext[i];
        if (a.Pnext2 != nullptr) pn += a.Pnext2[i];
        pn += acc[r];
        a.Pnext[i] = pn;
        sT[GK::kchunk(row, cl >> 2) + (cl & 3)] = pn;     // element (row, k = column)
    }
    __syncthreads();
    // ---- real Gram of the planar rows over this tile's 64 columns (combined to P P^H by c64_gram_combine_kernel) ----
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

// workspace: atom_fused_c64_real_count (atom_sweep.hpp) = Dhat 2KF + Alook~ 4K^2 + Ablk~ rows of block 0 (64 x 2K)
// + Aprev~ (2K x 64) + E~ (64 x 64)
inline bool atom_fused_c64_ok(int64_t F, int64_t K) { return atom_fused_c64_shape(F, K); }

// D_new (a copy of D on entry) <- the swept dictionary: the complex64 fast path.
inline int atom_sweep_fused_c64(dcp_handle* h, const c64* A, const c64* B, c64* Dnew, int64_t F64, int64_t K64,
                                AtomWs<c64>& w) {
    hipStream_t st = h->stream;
    const int K = (int)K64, F = (int)F64;
    const int nblk = K / 32, ntile = F / 64;
    if (w.fused_reals == nullptr) return fail(h, DCP_ERR_INTERNAL, "atom sweep: fused complex workspace missing");
    float* Dhat = w.fused_reals;
    float* Alook_ext = Dhat + (size_t)2 * K * F;
    float* Ablk0_ext = Alook_ext + (size_t)4 * K * K;
    float* Aprev_ext = Ablk0_ext + (size_t)64 * 2 * K;
    float* E_ext = Aprev_ext + (size_t)2 * K * 64;
    float* slabs = reinterpret_cast<float*>(w.slabs);            // 2 x slab_count floats
    float* Phat = reinterpret_cast<float*>(w.P);                 // 2 x [64, F] floats inside the complex buffer
    float* P2 = Phat + (size_t)2 * 64 * F;                       // a third [64, F]: second half of a split look-ahead
    if ((size_t)ntile * 64 * 64 > 2 * w.slab_count) return fail(h, DCP_ERR_INTERNAL, "atom slab plan");
    hipLaunchKernelGGL((atom_prep_kernel<c64>), dim3(grid_for((long)K * K, 256)), dim3(256), 0, st, 32, K, A,
                       w.Ablk, w.Wl, w.rden, w.Alook, w.Aprev, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    DCP_LAUNCH_OK(h, hipGetLastError());
    hipLaunchKernelGGL(c64_planar_rows_kernel<c64>, dim3(grid_for((long)K * F, 2048)), dim3(256), 0, st, (const c64*)Dnew,
                       (long)K, (long)F, Dhat);
    DCP_LAUNCH_OK(h, hipGetLastError());
    hipLaunchKernelGGL(c64_extend_left_kernel<c64>, dim3(grid_for((long)K * K, 256)), dim3(256), 0, st, (const c64*)w.Alook,
                       (long)K, (long)K, (long)K, Alook_ext, 2L * K);
    DCP_LAUNCH_OK(h, hipGetLastError());
    hipLaunchKernelGGL(c64_extend_left_kernel<c64>, dim3(grid_for(32L * K, 256)), dim3(256), 0, st, (const c64*)w.Ablk, 32L,
                       (long)K, (long)K, Ablk0_ext, 2L * K);
    DCP_LAUNCH_OK(h, hipGetLastError());
    hipLaunchKernelGGL(c64_extend_left_kernel<c64>, dim3(grid_for((long)K * 32, 256)), dim3(256), 0, st, (const c64*)w.Aprev,
                       (long)K, 32L, (long)kAtomBlkMax, Aprev_ext, 64L);
    DCP_LAUNCH_OK(h, hipGetLastError());
    constexpr int kLds = 65536;     // look-ahead panels (2 x 2 x 16 KiB) > the complex recursion's 40 KiB image
    {
        static DynLdsRaised raised;
        std::atomic<bool>& r = raised.on_current_device();
        if (!r) {
            DCP_LAUNCH_OK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&atom_recur_lookahead_c64_kernel<c64>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
            r = true;
        }
    }
    {   // block 0: planar P by the look-ahead product alone (full rows: Ablk~), its Gram matrix by the real core
        AtomFusedArgsC fa;
        fa.G = nullptr; fa.Wl = nullptr; fa.E = nullptr; fa.E_ext = nullptr;
        fa.Alook = Ablk0_ext; fa.Dhat = Dhat; fa.Bn = B; fa.Dold = Dnew; fa.rden = w.rden;
        fa.Pnext = Phat; fa.Pnext2 = nullptr; fa.K2 = 2 * K; fa.F = F; fa.has_next = 1; fa.ksplit = 1;
        hipLaunchKernelGGL(atom_recur_lookahead_c64_kernel<c64>, dim3(1 + ntile), dim3(kAtomRecurThreads), kLds, st, fa);
        DCP_LAUNCH_OK(h, hipGetLastError());
        GemmArgs<float> g;
        g.A = Phat; g.lda = F; g.B = Phat; g.ldb = F; g.M = 64; g.N = 64; g.K = F;
        g.tile = TILE_SMALL_DEEP;
        plan_splits<FORM_NT>(g, 64, 64, 4);
        if ((size_t)g.ksplits * 64 * 64 > 2 * w.slab_count) return fail(h, DCP_ERR_INTERNAL, "atom slab plan");
        DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, g, EpiSlab<float>{slabs, 64L, 64L * 64})));
        hipLaunchKernelGGL(c64_gram_combine_kernel<c64>, dim3(16), dim3(256), 0, st, (const float*)slabs, g.ksplits,
                           64L * 64, w.G);
        DCP_LAUNCH_OK(h, hipGetLastError());
    }
    for (int b = 0; b < nblk; ++b) {
        const int k0 = b * 32, k1 = k0 + 32;
        const bool has_next = (b + 1) < nblk;
        float* P = Phat + (size_t)(b & 1) * 64 * F;
        float* Pnext = Phat + (size_t)((b + 1) & 1) * 64 * F;
        AtomFusedArgsC fa;
        fa.G = w.G; fa.Wl = w.Wl + (long)k0 * kAtomBlkMax; fa.E = w.E; fa.E_ext = E_ext;
        fa.Alook = has_next ? Alook_ext + (size_t)(2 * k1) * (2 * K) : nullptr;
        fa.Dhat = Dhat;
        fa.Bn = has_next ? B + (long)k1 * F : nullptr;
        fa.Dold = has_next ? Dnew + (long)k1 * F : nullptr;
        fa.rden = has_next ? w.rden + k1 : nullptr;
        // the 2K-deep look-ahead product in two halves when that still fits one round of CUs: the launch is as long
        // as its slower part, and with F / 64 workgroups of 16 K blocks that was the product, not the recursion
        const int ksplit = (((2 * K) / 64) % 2 == 0 && 2 * ntile <= 256) ? 2 : 1;
        fa.Pnext = Pnext; fa.Pnext2 = P2; fa.K2 = 2 * K; fa.F = F; fa.has_next = has_next ? 1 : 0; fa.ksplit = ksplit;
        hipLaunchKernelGGL(atom_recur_lookahead_c64_kernel<c64>, dim3(has_next ? 1 + ksplit * ntile : 1), dim3(kAtomRecurThreads), kLds,
                           st, fa);
        DCP_LAUNCH_OK(h, hipGetLastError());
        AtomApplyArgsC aa;
        aa.E_ext = E_ext; aa.P = P; aa.Dblk = Dnew + (long)k0 * F; aa.Dhat_blk = Dhat + (size_t)(2 * k0) * F;
        aa.Aprev = has_next ? Aprev_ext + (size_t)(2 * k1) * 64 : nullptr;
        aa.Pnext = Pnext; aa.Pnext2 = (has_next && ksplit == 2) ? P2 : nullptr;
        aa.slabs = slabs; aa.F = F; aa.has_next = has_next ? 1 : 0;
        hipLaunchKernelGGL(atom_apply_c64_kernel<c64>, dim3(ntile), dim3(256), 0, st, aa);
        DCP_LAUNCH_OK(h, hipGetLastError());
        if (has_next) {
            hipLaunchKernelGGL(c64_gram_combine_tree_kernel<c64>, dim3(32), dim3(512), 0, st, (const float*)slabs,
                               ntile, 64L * 64, w.G);
            DCP_LAUNCH_OK(h, hipGetLastError());
        }
    }
    return DCP_OK;
}

}  // namespace dcp
