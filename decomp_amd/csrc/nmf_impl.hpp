// NMF multiplicative update on the device (templated on float / double).
//
// Reference path restated as kernels:
//   decomp/nmf_methods/batch_mu.py:8-26     outer loop, stop rule
//   decomp/nmf_methods/grads.py:77-93       x * max(pos,0) / max(neg,1e-15)
//   decomp/nmf_methods/grads.py:108-125     Gaussian gradient parts
//   decomp/nmf_methods/grads.py:143-160     Poisson gradient parts
//   decomp/utils/normalize.py:13-21         l2_strict
//
// l2 without a mask uses the Gram formulation:
//   (x D) D^T = x (D D^T)   and   x^T (x D) = (x^T x) D
// which removes the [N,F] intermediate and 2/3 of the flops; masked and KL updates
// cannot use it and keep the chained products with the [N,F] intermediate in HBM.
//
// One iteration =  stats(...)  [all-reduce of `stats` across ranks goes here]  update(...)
#pragma once
#include <math.h>

#include "gemm.hpp"
#include "handle.hpp"
#include "kernels_small.hpp"

namespace dcp {

inline int64_t nmf_stats_width(int64_t F, int64_t K, int likelihood, bool masked) {
    return (likelihood == DCP_LIK_L2 && !masked) ? (F + K) : 2 * F;
}

// Workgroups a split-K grid is sized for: ONE round of resident 128x128 workgroups
// (4 per CU x 256 CUs).  Measured: x^T Y at 16 splits (1024 WGs) 1.05 ms, at 30 (2 rounds) 1.15.
constexpr int kSplitTarget = 1024;
constexpr int kMaxSplits = 64;

template <class T>
struct NmfShape {
    int64_t N, F, K;
    int lik;
    bool masked;
};

// ---- workspace ---------------------------------------------------------------------
template <class T>
struct NmfStatsWs {
    T* Q = nullptr;       // [N,K]   negative part of the x gradient
    T* G = nullptr;       // [K,K]   D D^T (l2, no mask)
    T* f = nullptr;       // [N,F]   (x D) o M  or  KL ratio
    T* Ym = nullptr;      // [N,F]   Y o M when the caller did not pre-mask
    T* slabs = nullptr;   // split-K partials
    T* vecK = nullptr;    // [K]     KL: colsum(D) / colsum(x)
    T* part = nullptr;    // column-sum partials
    size_t slab_count = 0;
    // float + 0/1 mask: its row-bit image (mask_rowbits_kernel); the forward product then multiplies
    // by bits fetched ahead of its main loop instead of loading the float mask in its epilogue
    const uint32_t* mbits = nullptr;
};

// Split-K plan of the x-update GEMM (Y D^T, [N,F]x[K,F]): only when the row tiles alone
// cannot fill the chip (rows per GPU small, e.g. one shard of an 8-GPU run).
template <class T>
inline int nmf_xupdate_splits(int64_t N, int64_t F, int64_t K, GemmArgs<T>& pg) {
    pg.M = (int)N; pg.N = (int)K; pg.K = (int)F;
    pg.tile = TILE_AUTO;
    int bm = 64, bn = 64;
    if (std::is_same<T, float>::value) tier_dims(pick_tier<FORM_NT>(pg.M, pg.N, pg.K, TILE_AUTO, true), bm, bn);
    const long tiles = (long)ceil_div(N, bm) * ceil_div(K, bn);
    const long enough = (bm == 256) ? 192 : 512;
    if (tiles >= enough || F < 1024) {
        pg.ksplits = 1;
        return 1;
    }
    plan_splits<FORM_NT>(pg, (bm == 256) ? 256 : 1024, 8);
    return pg.ksplits;
}

template <class T>
inline size_t nmf_slab_elems(const NmfShape<T>& s) {
    // the largest split-K product of the step: stats GEMM [K, W] (and G [K,K])
    const int64_t W = nmf_stats_width(s.F, s.K, s.lik, s.masked);
    GemmArgs<T> a;
    a.M = (int)s.K; a.N = (int)W; a.K = (int)s.N;
    plan_splits<FORM_TN>(a, kSplitTarget, kMaxSplits);
    size_t stats_slabs = (size_t)a.ksplits * s.K * W;
    GemmArgs<T> g;
    g.M = (int)s.K; g.N = (int)s.K; g.K = (int)s.F;
    plan_splits<FORM_NT>(g, 512, kMaxSplits, 16);
    size_t g_slabs = (size_t)g.ksplits * s.K * s.K;
    GemmArgs<T> pg;
    const int ps = nmf_xupdate_splits<T>(s.N, s.F, s.K, pg);
    size_t x_slabs = ps > 1 ? (size_t)ps * s.N * s.K : 0;
    if (ps > 1 && s.masked && s.lik == DCP_LIK_L2) {      // the stacked [f ; Y o M] . D^T launch
        GemmArgs<T> sg;
        const int ss = nmf_xupdate_splits<T>(2 * s.N, s.F, s.K, sg);
        const size_t st_slabs = (size_t)(ss > 1 ? ss : 1) * 2 * s.N * s.K;
        if (st_slabs > x_slabs) x_slabs = st_slabs;
    }
    size_t m = stats_slabs > g_slabs ? stats_slabs : g_slabs;
    return m > x_slabs ? m : x_slabs;
}

template <class T>
inline void nmf_plan_stats(WsPlan& plan, const NmfShape<T>& s, bool need_ym) {
    const bool gram = (s.lik == DCP_LIK_L2 && !s.masked);
    plan.add<T>((size_t)s.N * s.K);                       // Q
    if (gram) plan.add<T>((size_t)s.K * s.K);             // G
    if (!gram) plan.add<T>((size_t)s.N * s.F);            // f
    if (need_ym) plan.add<T>((size_t)s.N * s.F);          // Ym
    plan.add<T>(nmf_slab_elems(s));                       // slabs
    plan.add<T>((size_t)s.K);                             // vecK
    plan.add<T>((size_t)64 * (s.K > s.F ? s.K : s.F));    // part
}

template <class T>
inline int nmf_carve_stats(dcp_handle* h, NmfStatsWs<T>& w, const NmfShape<T>& s, bool need_ym) {
    const bool gram = (s.lik == DCP_LIK_L2 && !s.masked);
    w.Q = ws_alloc<T>(h, (size_t)s.N * s.K);
    if (gram) w.G = ws_alloc<T>(h, (size_t)s.K * s.K);
    if (!gram) w.f = ws_alloc<T>(h, (size_t)s.N * s.F);
    if (need_ym) w.Ym = ws_alloc<T>(h, (size_t)s.N * s.F);
    w.slab_count = nmf_slab_elems(s);
    w.slabs = ws_alloc<T>(h, w.slab_count);
    w.vecK = ws_alloc<T>(h, (size_t)s.K);
    w.part = ws_alloc<T>(h, (size_t)64 * (s.K > s.F ? s.K : s.F));
    if (!w.Q || !w.slabs || !w.vecK || !w.part || (gram && !w.G) || (!gram && !w.f) ||
        (need_ym && !w.Ym))
        return fail(h, DCP_ERR_INTERNAL, "nmf workspace plan mismatch");
    return DCP_OK;
}

template <class T>
struct NmfUpdateWs {
    T* U = nullptr;        // [K,F] un-normalised new dictionary
    T* rowmax = nullptr;   // [K]
};
template <class T>
inline void nmf_plan_update(WsPlan& plan, int64_t F, int64_t K) {
    plan.add<T>((size_t)K * F);
    plan.add<T>((size_t)K);
}
template <class T>
inline int nmf_carve_update(dcp_handle* h, NmfUpdateWs<T>& w, int64_t F, int64_t K) {
    w.U = ws_alloc<T>(h, (size_t)K * F);
    w.rowmax = ws_alloc<T>(h, (size_t)K);
    if (!w.U || !w.rowmax) return fail(h, DCP_ERR_INTERNAL, "nmf workspace plan mismatch");
    return DCP_OK;
}

#define DCP_LAUNCH_OK(h, what)                                                        \
    do {                                                                              \
        hipError_t _e = (what);                                                       \
        if (_e != hipSuccess)                                                         \
            return dcp::fail((h), DCP_ERR_HIP, std::string("launch failed: ") +       \
                                                   hipGetErrorString(_e));            \
    } while (0)

// out[K] = column sums of a[rows, cols] (deterministic two-stage).
template <class T>
inline int column_sums(dcp_handle* h, const T* a, long ld, long rows, long cols, T* part, T* out) {
    const int stripes = rows >= 64 ? 64 : (rows > 0 ? (int)rows : 1);
    const long rows_per = (rows + stripes - 1) / stripes;
    dim3 grid(grid_for(cols, 64), stripes);
    hipLaunchKernelGGL((colsum_partial_kernel<T>), grid, dim3(256), 0, h->stream, a, ld, rows,
                       cols, rows_per, part);
    DCP_LAUNCH_OK(h, hipGetLastError());
    hipLaunchKernelGGL((reduce_slabs_kernel<T>), dim3(grid_for(cols, 64)), dim3(256), 0, h->stream,
                       part, cols, stripes, cols, out);
    DCP_LAUNCH_OK(h, hipGetLastError());
    return DCP_OK;
}

// ---- x update + local statistics ------------------------------------------------------
// Ypre: Y already multiplied by the mask (or Y itself when there is no mask).
template <class T>
inline int nmf_stats(dcp_handle* h, const T* Ypre, const T* mask, const T* Xin, T* Xout, const T* D,
                     const NmfShape<T>& s, T* stats, NmfStatsWs<T>& w, int phases = 3) {
    // phases: bit 0 = the x update (Xin -> Xout), bit 1 = the D-side sums (with Xout, or with
    // Xin when the x update is skipped)
    hipStream_t st = h->stream;
    const int N = (int)s.N, F = (int)s.F, K = (int)s.K;
    const bool gram = (s.lik == DCP_LIK_L2 && !s.masked);
    const int W = (int)nmf_stats_width(s.F, s.K, s.lik, s.masked);

    // forward product x.D with an elementwise epilogue into the [N,F] intermediate w.f
    auto forward = [&](const T* X) -> int {
        ProfScope ps(h, DCP_PROF_FWD);
        GemmArgs<T> fa;
        fa.A = X; fa.lda = K; fa.B = D; fa.ldb = F; fa.M = N; fa.N = F; fa.K = K;
        if (s.lik == DCP_LIK_L2) {  // f = (x D) o M                 (grads.py:113,123)
            if constexpr (std::is_same<T, float>::value) {
                if (w.mbits != nullptr) {
                    DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, fa, EpiMulMaskBits{w.mbits, (long)F, w.f, (long)F})));
                    return DCP_OK;
                }
            }
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, fa, EpiMulMask<T>{mask, F, w.f, F})));
        } else                       // r = (Y o M) / (x D + eps)     (grads.py:145-149)
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, fa, EpiKlRatio<T>{Ypre, F, nullptr, 0, w.f, F})));
        return DCP_OK;
    };

    const T* X = Xin;
    bool x_done = false;
    if (phases & 1) {
    // ---------------- x <- x * max(pos,0) / max(neg,eps) ----------------
    const T* xnum_A = Ypre;   // left operand of the positive-part GEMM (. D^T)
    const T* xden = w.Q;      // negative part
    long ld_xden = K;
    // Few rows per GPU (one shard of a multi-GPU run): Y.D^T splits its F reduction so that all CUs work, and
    // the quotient moves into the 16-byte epilogue of the x.G product, which then runs AFTER it (float32).
    // Not for an in-place update: that product reads whole rows of x while other tiles write them.
    GemmArgs<T> pg_probe;
    const bool split_gram = gram && std::is_same<T, float>::value && Xout != Xin &&
                            nmf_xupdate_splits<T>(s.N, s.F, s.K, pg_probe) > 1;
    if (gram) {
        {   // G = D D^T  (split over F, partial slabs summed in order)
            ProfScope ps(h, DCP_PROF_GRAM);
            GemmArgs<T> g;
            g.A = D; g.lda = F; g.B = D; g.ldb = F; g.M = K; g.N = K; g.K = F;
            if (!std::is_same<T, float>::value) g.tile = TILE_SMALL_DEEP;   // float64: 64 x 64 tiles (see f64_tier)
            plan_splits<FORM_NT>(g, 512, kMaxSplits, 16);   // tiny output: 256-deep splits (measured best)
            DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, g, EpiSlab<T>{w.slabs, K, (long)K * K})));
            hipLaunchKernelGGL((reduce_slabs_kernel<T>), dim3(grid_for((long)K * K)), dim3(256), 0,
                               st, w.slabs, (long)K * K, g.ksplits, (long)K * K, w.G);
            DCP_LAUNCH_OK(h, hipGetLastError());
        }
        if (!split_gram) {   // Q = x G
            ProfScope ps(h, DCP_PROF_XNEG);
            GemmArgs<T> q;
            q.A = Xin; q.lda = K; q.B = w.G; q.ldb = K; q.M = N; q.N = K; q.K = K;
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, q, EpiStore<T>{w.Q, K})));
        }
    } else if (s.lik == DCP_LIK_L2) {
        DCP_TRY(forward(Xin));
        if constexpr (std::is_same<T, float>::value) {
            // masked l2 with few rows per GPU: both parts of the x gradient are products against the
            // same D^T -- [f ; Y o M] . D^T as ONE split-K launch over the stacked rows (2N x K output:
            // twice the tiles, half the splits, one set of fixed costs), then one quotient kernel
            GemmArgs<T> sg;
            sg.A = w.f; sg.lda = F; sg.A2 = Ypre; sg.lda2 = F; sg.m_a1 = N;
            sg.B = D; sg.ldb = F; sg.M = 2 * N; sg.N = K; sg.K = F;
            GemmArgs<T> probe;
            const bool split_single = nmf_xupdate_splits<T>(s.N, s.F, s.K, probe) > 1;
            if (split_single && (N % 256) == 0 && (F % 16) == 0) {
                ProfScope ps(h, DCP_PROF_XUPDATE);
                nmf_xupdate_splits<T>(2 * s.N, s.F, s.K, sg);      // plan for the stacked problem
                sg.A = w.f; sg.lda = F; sg.A2 = Ypre; sg.lda2 = F; sg.m_a1 = N;
                sg.B = D; sg.ldb = F;
                if (sg.ksplits < 1) sg.ksplits = 1;
                if ((size_t)sg.ksplits * 2 * N * K > w.slab_count)
                    return fail(h, DCP_ERR_INTERNAL, "nmf stacked x-update slab plan mismatch");
                DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, sg, EpiSlab<T>{w.slabs, K, (long)2 * N * K})));
                hipLaunchKernelGGL((mu_quotient_stacked_kernel<T>), dim3(grid_for((long)N * K)), dim3(256), 0,
                                   st, Xin, (const T*)w.slabs, (long)2 * N * K, sg.ksplits, (long)N * K, Xout);
                DCP_LAUNCH_OK(h, hipGetLastError());
                x_done = true;
            }
        }
        if (!x_done) {
            ProfScope ps(h, DCP_PROF_XNEG);   // Q = f D^T
            GemmArgs<T> q;
            q.A = w.f; q.lda = F; q.B = D; q.ldb = F; q.M = N; q.N = K; q.K = F;
            DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, q, EpiStore<T>{w.Q, K})));
        }
    } else {
        DCP_TRY(forward(Xin));
        xnum_A = w.f;
        ProfScope ps(h, DCP_PROF_XNEG);
        if (!s.masked) {   // neg = colsum(D), one value per column of x (grads.py:146)
            hipLaunchKernelGGL((rowsum_kernel<T>), dim3(K), dim3(256), 0, st, D, (long)F, (long)F,
                               w.vecK);
            DCP_LAUNCH_OK(h, hipGetLastError());
            xden = w.vecK;
            ld_xden = 0;
        } else {           // neg = M D^T (grads.py:150)
            GemmArgs<T> q;
            q.A = mask; q.lda = F; q.B = D; q.ldb = F; q.M = N; q.N = K; q.K = F;
            DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, q, EpiStore<T>{w.Q, K})));
        }
    }
    if (!x_done) {   // x <- x * max(pos, 0) / max(neg, eps)
        ProfScope ps(h, DCP_PROF_XUPDATE);
        GemmArgs<T> pg;
        pg.A = xnum_A; pg.lda = F; pg.B = D; pg.ldb = F; pg.M = N; pg.N = K; pg.K = F;
        const int psplits = nmf_xupdate_splits<T>(s.N, s.F, s.K, pg);
        if (psplits <= 1) {
            // enough row tiles to fill the chip: quotient fused into the GEMM epilogue
            pg.ksplits = 1;
            DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, pg, EpiMuNum<T>{Xin, K, xden, ld_xden, Xout, K})));
        } else {
            // few rows per GPU (a shard of a multi-GPU run): split the F reduction so that all
            // CUs work, then sum the slabs in order inside the quotient kernel
            if ((size_t)pg.ksplits * N * K > w.slab_count)
                return fail(h, DCP_ERR_INTERNAL, "nmf x-update slab plan mismatch");
            DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, pg, EpiSlab<T>{w.slabs, K, (long)N * K})));
            if (split_gram) {
                GemmArgs<T> q;   // x_new = x * max(sum slabs, 0) / max(x G, eps)
                q.A = Xin; q.lda = K; q.B = w.G; q.ldb = K; q.M = N; q.N = K; q.K = K;
                DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, q, EpiMuDenSlabs<T>{Xin, K, w.slabs, K, (long)N * K,
                                                                        pg.ksplits, Xout, K})));
            } else {
                hipLaunchKernelGGL((mu_quotient_slabs_kernel<T>), dim3(grid_for((long)N * K)), dim3(256),
                                   0, st, Xin, (const T*)w.slabs, (long)N * K, pg.ksplits, xden,
                                   (long)ld_xden, (long)N, (long)K, Xout);
                DCP_LAUNCH_OK(h, hipGetLastError());
            }
        }
    }
    X = Xout;
    }
    if (!(phases & 2)) return DCP_OK;

    // ---------------- local D-side sums with the NEW x ----------------
    GemmArgs<T> sa;
    sa.A = X; sa.lda = K; sa.M = K; sa.N = W; sa.K = N;
    if (gram) {
        sa.B = Ypre; sa.ldb = F; sa.B2 = X; sa.ldb2 = K; sa.n_b1 = F;      // [ x^T Y | x^T x ]
    } else if (s.lik == DCP_LIK_L2) {
        DCP_TRY(forward(X));
        sa.B = Ypre; sa.ldb = F; sa.B2 = w.f; sa.ldb2 = F; sa.n_b1 = F;    // [ x^T Ym | x^T f ]
    } else {
        DCP_TRY(forward(X));
        if (s.masked) {
            sa.B = w.f; sa.ldb = F; sa.B2 = mask; sa.ldb2 = F; sa.n_b1 = F;  // [ x^T r | x^T M ]
        } else {
            sa.B = w.f; sa.ldb = F; sa.N = F;                               // x^T r only
        }
    }
    const int Wg = sa.N;  // width produced by the GEMM (KL no-mask: F, the rest is filled below)
    plan_splits<FORM_TN>(sa, kSplitTarget, kMaxSplits);
    if ((size_t)sa.ksplits * K * Wg > w.slab_count)
        return fail(h, DCP_ERR_INTERNAL, "nmf slab plan mismatch");
    {
        ProfScope ps(h, DCP_PROF_STATS);
        DCP_LAUNCH_OK(h, (gemm<FORM_TN>(st, sa, EpiSlab<T>{w.slabs, Wg, (long)K * Wg})));
    }
    ProfScope ps(h, DCP_PROF_STATS_SUM);
    if (Wg == W) {
        launch_reduce_slabs<T>(st, w.slabs, (long)K * W, sa.ksplits, (long)K * W, stats);
        DCP_LAUNCH_OK(h, hipGetLastError());
    } else {
        // KL without mask: numerator from the GEMM, denominator = colsum(x) broadcast over F
        // (grads.py:155: x.T.sum(axis=1, keepdims=True)).  Summed over ranks like the rest.
        hipLaunchKernelGGL((reduce_slabs_rows_kernel<T>), dim3(grid_for((long)K * F)), dim3(256), 0,
                           st, w.slabs, (long)K * F, sa.ksplits, (long)K, (long)F, stats, (long)W);
        DCP_LAUNCH_OK(h, hipGetLastError());
        DCP_TRY(column_sums<T>(h, X, K, N, K, w.part, w.vecK));
        hipLaunchKernelGGL((bcast_rows_kernel<T>), dim3(grid_for((long)K * F)), dim3(256), 0, st,
                           w.vecK, (long)K, (long)F, stats + F, (long)W);
        DCP_LAUNCH_OK(h, hipGetLastError());
    }
    return DCP_OK;
}

// ---- the two parts of the x gradient as explicit arrays (the reference's plugin surface) ------
// Gaussian.grad_x / Poisson.grad_x (grads.py:108-115, 143-150): pos, neg [N, K].
//   l2        : pos = (Y o M) D^T,  neg = ((x D) o M) D^T   (no mask: neg = x (D D^T), the Gram identity)
//   kl        : pos = ((Y o M) / (x D + eps)) D^T,  neg = colsum(D) on every row (no mask) or M D^T
// (for kl without a mask the reference returns the [1, K] row d.T.sum(axis=0, keepdims=True); the caller
//  slices row 0 of neg).  Ypre = Y o M (or Y).
template <class T>
inline int nmf_grad_x(dcp_handle* h, const T* Ypre, const T* mask, const T* X, const T* D,
                      const NmfShape<T>& s, T* pos, T* neg, NmfStatsWs<T>& w) {
    hipStream_t st = h->stream;
    const int N = (int)s.N, F = (int)s.F, K = (int)s.K;
    const bool gram = (s.lik == DCP_LIK_L2 && !s.masked);
    const T* pos_A = Ypre;
    if (gram) {
        GemmArgs<T> g;
        g.A = D; g.lda = F; g.B = D; g.ldb = F; g.M = K; g.N = K; g.K = F;
        plan_splits<FORM_NT>(g, 512, kMaxSplits, 16);
        DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, g, EpiSlab<T>{w.slabs, K, (long)K * K})));
        hipLaunchKernelGGL((reduce_slabs_kernel<T>), dim3(grid_for((long)K * K)), dim3(256), 0, st, w.slabs,
                           (long)K * K, g.ksplits, (long)K * K, w.G);
        DCP_LAUNCH_OK(h, hipGetLastError());
        GemmArgs<T> q;
        q.A = X; q.lda = K; q.B = w.G; q.ldb = K; q.M = N; q.N = K; q.K = K;
        DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, q, EpiStore<T>{neg, K})));
    } else {
        GemmArgs<T> fa;
        fa.A = X; fa.lda = K; fa.B = D; fa.ldb = F; fa.M = N; fa.N = F; fa.K = K;
        if (s.lik == DCP_LIK_L2) {
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, fa, EpiMulMask<T>{mask, F, w.f, F})));
            GemmArgs<T> q;
            q.A = w.f; q.lda = F; q.B = D; q.ldb = F; q.M = N; q.N = K; q.K = F;
            DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, q, EpiStore<T>{neg, K})));
        } else {
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, fa, EpiKlRatio<T>{Ypre, F, nullptr, 0, w.f, F})));
            pos_A = w.f;
            if (!s.masked) {
                hipLaunchKernelGGL((rowsum_kernel<T>), dim3(K), dim3(256), 0, st, D, (long)F, (long)F, w.vecK);
                DCP_LAUNCH_OK(h, hipGetLastError());
                // neg[n, k] = colsum(D)[k] for every n: bcast_rows_kernel writes out[r, c] = v[r], so build the
                // transposed broadcast with the quotient kernel's column mode instead: one pass, den_bcast = 2
                hipLaunchKernelGGL((bcast_cols_kernel<T>), dim3(grid_for((long)N * K)), dim3(256), 0, st,
                                   (const T*)w.vecK, (long)N, (long)K, neg);
                DCP_LAUNCH_OK(h, hipGetLastError());
            } else {
                GemmArgs<T> q;
                q.A = mask; q.lda = F; q.B = D; q.ldb = F; q.M = N; q.N = K; q.K = F;
                DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, q, EpiStore<T>{neg, K})));
            }
        }
    }
    GemmArgs<T> pg;
    pg.A = pos_A; pg.lda = F; pg.B = D; pg.ldb = F; pg.M = N; pg.N = K; pg.K = F;
    DCP_LAUNCH_OK(h, (gemm<FORM_NT>(st, pg, EpiStore<T>{pos, K})));
    return DCP_OK;
}

// ---- D update from the (all-reduced) statistics ---------------------------------------
template <class T>
inline int nmf_update(dcp_handle* h, const T* stats, const T* D, T* D_new, int64_t F64, int64_t K64,
                      int lik, bool masked, T* maxdiff_dev, NmfUpdateWs<T>& w,
                      T* maxdiff_next = nullptr, unsigned int* ticket = nullptr, T* host_out = nullptr) {
    hipStream_t st = h->stream;
    const int F = (int)F64, K = (int)K64;
    const bool gram = (lik == DCP_LIK_L2 && !masked);
    const int W = (int)nmf_stats_width(F64, K64, lik, masked);
    {
        ProfScope ps(h, DCP_PROF_DUPDATE);
        if (gram) {
            // U = D * max(x^T Y, 0) / max((x^T x) D, eps): quotient fused into the S.D GEMM
            GemmArgs<T> a;
            a.A = stats + F; a.lda = W; a.B = D; a.ldb = F; a.M = K; a.N = F; a.K = K;
            // float: K = 256 deep and latency bound: 64-deep K blocks on the small tile.  double: the
            // default tier (the fp64 MFMA core for >= 128 x 128 outputs; the small tile would send it
            // to the generic VALU core: 72 us against the float path's 16 us)
            if (std::is_same<T, float>::value) a.tile = (K % 64 == 0) ? TILE_SMALL_DEEP : TILE_SMALL;
            else a.tile = TILE_AUTO;
            DCP_LAUNCH_OK(h, (gemm<FORM_NN>(st, a, EpiMuDen<T>{D, F, stats, W, w.U, F})));
        } else {
            hipLaunchKernelGGL((mu_quotient_kernel<T>), dim3(grid_for((long)K * F)), dim3(256), 0,
                               st, D, (long)F, stats, (long)W, stats + F, (long)W, 0, (long)K,
                               (long)F, w.U, (long)F);
            DCP_LAUNCH_OK(h, hipGetLastError());
        }
    }
    // D_new = l2_strict(U) ; max |D - D_new|
    ProfScope ps(h, DCP_PROF_DNORM);
    if (maxdiff_next != nullptr) {
        // ping-pong slots: *maxdiff_dev is zero on entry (cleared by the previous iteration),
        // the max is formed with one atomic per row, and the other slot is cleared for the next
        hipLaunchKernelGGL((row_normalize_kernel<T>), dim3(K), dim3(256), 0, st, w.U, (long)F,
                           (long)F, 1, D, (long)F, D_new, (long)F, (T*)nullptr, (T*)nullptr,
                           maxdiff_dev, maxdiff_next, ticket, host_out);
        DCP_LAUNCH_OK(h, hipGetLastError());
        return DCP_OK;
    }
    hipLaunchKernelGGL((row_normalize_kernel<T>), dim3(K), dim3(256), 0, st, w.U, (long)F, (long)F,
                       1, D, (long)F, D_new, (long)F, w.rowmax, (T*)nullptr, (T*)nullptr,
                       (T*)nullptr);
    DCP_LAUNCH_OK(h, hipGetLastError());
    hipLaunchKernelGGL((final_max_kernel<T>), dim3(1), dim3(256), 0, st, w.rowmax, (long)K,
                       maxdiff_dev);
    DCP_LAUNCH_OK(h, hipGetLastError());
    return DCP_OK;
}

// ---- residual ||(Y - X D) o M||_F ---------------------------------------------------------
template <class T>
inline int nmf_residual(dcp_handle* h, const T* Y, const T* mask, const T* X, const T* D, int64_t N,
                        int64_t F, int64_t K, T* tmpNF, double* partial_dev, int nblocks) {
    GemmArgs<T> a;
    a.A = X; a.lda = K; a.B = D; a.ldb = F; a.M = (int)N; a.N = (int)F; a.K = (int)K;
    DCP_LAUNCH_OK(h, (gemm<FORM_NN>(h->stream, a, EpiResidual<T>{Y, F, mask, F, tmpNF, F})));
    hipLaunchKernelGGL((sumsq_partial_kernel<T>), dim3(nblocks), dim3(256), 0, h->stream, tmpNF,
                       (long)N * F, partial_dev);
    DCP_LAUNCH_OK(h, hipGetLastError());
    return DCP_OK;
}

}  // namespace dcp

namespace dcp {

// Loop-invariant part of a masked solve: Ym = Y o M (grads.py:114,124 recompute it every call) and,
// for float, the row-bit image of a 0/1 mask.  *binary (host) tells whether `bits` may stand in for
// the mask.  Synchronises the stream once (the flag read-back).
template <class T>
inline int nmf_mask_prepare(dcp_handle* h, const T* Y, const T* mask, int64_t N, int64_t F, T* Ym,
                            uint32_t* bits, int* flag_dev, int* binary) {
    hipLaunchKernelGGL((mul_mask_kernel<T>), dim3(grid_for(N * F)), dim3(256), 0, h->stream, Y, mask,
                       (long)N, (long)F, (long)F, Ym);
    DCP_LAUNCH_OK(h, hipGetLastError());
    *binary = 0;
    if (!std::is_same<T, float>::value || bits == nullptr) return DCP_OK;
    void* hostv = nullptr;
    DCP_TRY(host_scratch(h, 64, &hostv));
    DCP_HIP_OK(h, hipMemsetAsync(flag_dev, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL((mask_rowbits_kernel<T>), dim3(grid_for(((N + 31) / 32) * F)), dim3(256), 0,
                       h->stream, mask, (long)N, (long)F, bits, flag_dev);
    DCP_LAUNCH_OK(h, hipGetLastError());
    DCP_HIP_OK(h, hipMemcpyAsync(hostv, flag_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    DCP_HIP_OK(h, hipStreamSynchronize(h->stream));
    *binary = (*reinterpret_cast<int*>(hostv) == 0) ? 1 : 0;
    return DCP_OK;
}

inline size_t mask_bits_words(int64_t N, int64_t F) { return (size_t)((N + 31) / 32) * (size_t)F; }

}  // namespace dcp
