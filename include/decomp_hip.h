/*
 * decomp_hip.h -- C ABI of libdecomp_hip.so, the MI355X (gfx950) implementation of
 * deComP's iterative-update hot path.
 *
 * The reference (fujii-team/deComP) has no FFI: its "device layer" is the NumPy/CuPy
 * array-module handle `xp` (decomp/utils/cp_compat.py:9-24) threaded through every
 * solver.  This header is what replaces `xp` + the L1/L2 solver bodies for the hot
 * path; each entry point cites the reference function it stands in for.  The
 * reference-side binding (a ctypes stub) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: opaque handle, raw DEVICE pointers, sizes; no C++ or torch types.
 *   - every function returns an int status (DCP_OK = 0, negative = error); no C++
 *     exception crosses the ABI.  dcp_last_error_string() gives the message.
 *   - arrays are C-contiguous row-major, exactly as the reference's NumPy arrays:
 *       y[N,F]  x[N,K]  D[K,F]  mask[N,F]   (N samples, F features/channels, K atoms)
 *   - dtype suffixes: f32, f64 (real), c64, c128 (interleaved re,im pairs).
 *   - work is enqueued on the handle's stream (dcp_set_stream); functions that
 *     return host scalars (iteration counts, max|dD|) synchronise that stream
 *     before returning, the *_async step functions do not.
 *   - a handle is bound to one device and is not re-entrant.
 */
#ifndef DECOMP_HIP_H
#define DECOMP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dcp_handle dcp_handle;

enum {
    DCP_OK = 0,
    DCP_ERR_INVALID = -1,   /* bad argument (null pointer, negative size, bad enum) */
    DCP_ERR_HIP = -2,       /* a HIP runtime call failed */
    DCP_ERR_NOMEM = -3,     /* workspace allocation failed */
    DCP_ERR_INTERNAL = -4,  /* library bug (workspace plan mismatch, ...) */
    DCP_ERR_UNSUPPORTED = -5,
    DCP_ERR_REF_TYPEERROR = -6, /* the reference raises TypeError on this input (lasso.py:509) */
    DCP_ERR_COMM = -7       /* librccl missing, no communicator on the handle, or an RCCL call failed */
};

/* likelihood codes: decomp/nmf_methods/grads.py:7-14 */
enum { DCP_LIK_L2 = 0, DCP_LIK_KL = 1 };

/* LASSO solver codes: decomp/lasso.py:13 */
enum { DCP_LASSO_ISTA = 0, DCP_LASSO_ACC_ISTA = 1, DCP_LASSO_FISTA = 2, DCP_LASSO_CD = 3,
       DCP_LASSO_PARALLEL_CD = 4, DCP_LASSO_ADMM = 5 };
/* dcp_dict_*: OR this into lasso_method for the '_pos' (non-negative) solvers */
enum { DCP_LASSO_POSITIVE = 0x100 };

/* ---- lifetime ----------------------------------------------------------------- */
int dcp_create(dcp_handle** out, int device);
int dcp_destroy(dcp_handle* h);
/* hipStream_t passed as void*; NULL = the device's default stream.  Cheap (no HIP call): the library's
 * workspace is ordered between the old and the new stream by the next call that uses workspace (an event
 * recorded on the OLD stream and waited for on the new one, no host synchronisation); the row movers use none
 * and are never ordered against other streams' work.
 * LIFETIME (hard rule): a stream handed to the handle must stay alive until the next workspace-using call has
 * been issued through this handle on another stream (that call still records an event on the old stream), or
 * until the handle is destroyed.  Destroying it earlier is a use-after-free inside the HIP runtime, not an
 * error the library can detect or recover from. */
int dcp_set_stream(dcp_handle* h, void* hip_stream);
const char* dcp_last_error_string(dcp_handle* h);
/* compile-time facts, for the loader's sanity check */
const char* dcp_build_info(void);

/* ---- multi-GPU: the handle's RCCL communicator ---------------------------------- */
/* The reference has no multi-device path; the data-parallel form of its loops (rows of y / x / mask
 * sharded over one process per GPU, D replicated) needs exactly ONE exchange per outer iteration: the
 * sum over ranks of the D-side statistics (grads.py:117-125 / dictionary_learning.py:147-152 are sums over
 * rows).  That all-reduce runs INSIDE the library, on the handle's own stream, through RCCL over xGMI:
 *   dcp_comm_unique_id : rank 0 draws the 128-byte id (ncclGetUniqueId) into HOST memory; the launcher's
 *                        own channel (decomp_amd.sharded: torch.distributed's store) carries it to the
 *                        other ranks -- the only thing that channel is used for.
 *   dcp_comm_init      : collective over all ranks (ncclCommInitRank on the handle's device).
 *   dcp_comm_allreduce_sum_* : in-place sum of buf[count] over the ranks, asynchronous on the handle's
 *                        stream (complex data: pass the interleaved (re, im) floats, count doubled).
 * librccl is bound at run time (the copy the process already holds, else the system one); without it
 * these calls return DCP_ERR_COMM and everything else of the library works.
 *   dcp_comm_set_external : instead of RCCL, the caller's own exchange (MPI, gloo, a test double): fn(buf, count,
 *                        dtype (0 = float32, 1 = float64), hip_stream, user) must leave the element-wise sum over
 *                        all ranks in the DEVICE array buf[count], ordered before anything enqueued on hip_stream
 *                        afterwards (e.g. synchronise the stream, exchange through the host, copy back on the same
 *                        stream), and return 0.  The sharded loops call it exactly where they would call
 *                        ncclAllReduce; decomp_amd.sharded uses it for process groups RCCL cannot serve (gloo with
 *                        several ranks on one GPU), so the SAME in-library loop runs in the two-rank tests.
 *   dcp_memcpy         : hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault) on the handle's stream (what such a
 *                        callback needs to stage through pinned host memory). */
enum { DCP_COMM_ID_BYTES = 128 };
typedef int (*dcp_allreduce_fn)(void* buf, int64_t count, int dtype, void* hip_stream, void* user);
int dcp_comm_set_external(dcp_handle* h, dcp_allreduce_fn fn, void* user, int rank, int world);
int dcp_memcpy(dcp_handle* h, void* dst, const void* src, int64_t bytes);
int dcp_comm_unique_id(void* id_out, int64_t id_bytes);
int dcp_comm_init(dcp_handle* h, const void* unique_id, int rank, int world);
int dcp_comm_destroy(dcp_handle* h);
/* *world = 0 when the handle has no communicator */
int dcp_comm_info(dcp_handle* h, int* rank, int* world);
int dcp_comm_allreduce_sum_f32(dcp_handle* h, float* buf, int64_t count);
int dcp_comm_allreduce_sum_f64(dcp_handle* h, double* buf, int64_t count);

/* ---- per-kernel timing (measurement aid, used by bench.py) ---------------------- */
/* While enabled, every labelled kernel group of the solvers is bracketed by hipEvents on
 * the handle's stream.  dcp_profile_read synchronises the stream, folds the pending
 * event pairs into per-label totals and returns total milliseconds and launch count. */
enum {
    DCP_PROF_GRAM = 0,        /* D D^T (split-K) + slab sum                      */
    DCP_PROF_XNEG = 1,        /* negative part of the x gradient: x G, f D^T, M D^T */
    DCP_PROF_XUPDATE = 2,     /* Y D^T GEMM with the fused MU quotient epilogue  */
    DCP_PROF_FWD = 3,         /* (x D) o M / KL ratio: the [N,F] intermediate     */
    DCP_PROF_STATS = 4,       /* x^T [Y | x] split-K GEMM                         */
    DCP_PROF_STATS_SUM = 5,   /* slab sum of the statistics                       */
    DCP_PROF_DUPDATE = 6,     /* (x^T x) D GEMM + quotient (or elementwise quotient) */
    DCP_PROF_DNORM = 7,       /* l2_strict + max|dD|                              */
    DCP_PROF_MISC = 8,
    DCP_PROF_EXCHANGE = 9,    /* the all-reduce of the statistics (sharded loops)  */
    DCP_PROF_NLABELS = 10
};
int dcp_profile_enable(dcp_handle* h, int on);
/* restrict the brackets to the labels whose bit is set (each bracket costs ~4 us of stream time:
 * bench.py times only the dominant kernel inside its timed steps) */
int dcp_profile_select(dcp_handle* h, unsigned label_mask);
int dcp_profile_reset(dcp_handle* h);
int dcp_profile_read(dcp_handle* h, int label, double* total_ms, int64_t* count);
const char* dcp_profile_label_name(int label);

/* ---- utilities ---------------------------------------------------------------- */
/* utils/normalize.py:2-21.  Rows of U[K,F] divided by sqrt(sum|u|^2) (strict != 0)
 * or by sqrt(max(sum|u|^2, 1)) (strict == 0), in place. */
int dcp_l2_normalize_f32(dcp_handle* h, float* U, int64_t K, int64_t F, int strict);
int dcp_l2_normalize_f64(dcp_handle* h, double* U, int64_t K, int64_t F, int strict);
int dcp_l2_normalize_c64(dcp_handle* h, void* U, int64_t K, int64_t F, int strict);
int dcp_l2_normalize_c128(dcp_handle* h, void* U, int64_t K, int64_t F, int strict);

/* utils/assertion.py:95-100 (assert_nonnegative): number of elements for which
 * `x >= 0` is false (negative values and NaNs), written to the HOST integer *count. */
int dcp_count_negative_f32(dcp_handle* h, const float* x, int64_t n, int64_t* count);
int dcp_count_negative_f64(dcp_handle* h, const double* x, int64_t n, int64_t* count);

/* math_utils/eigen.py:9-20 (spectral_radius_Gershgorin): for a batch X[batch, n, n],
 * out[b] = max_j sum_i |X[b, i, j]| (DEVICE array of the real dtype, length batch).
 * Asynchronous on the handle's stream. */
int dcp_gershgorin_f32(dcp_handle* h, const float* X, int64_t batch, int64_t n, float* out);
int dcp_gershgorin_f64(dcp_handle* h, const double* X, int64_t batch, int64_t n, double* out);
int dcp_gershgorin_c64(dcp_handle* h, const void* X, int64_t batch, int64_t n, float* out);
int dcp_gershgorin_c128(dcp_handle* h, const void* X, int64_t batch, int64_t n, double* out);

/* math_utils/linalg.py:9-38 (inv, "batch version of np.linalg.inv"): out[b] = X[b]^-1 for a batch
 * X[batch, n, n]; Gauss-Jordan with partial pivoting, computed in double precision (complex double) whatever
 * the storage dtype, one workgroup per matrix.  A singular matrix yields inf / nan entries (NumPy raises
 * LinAlgError).  X and out must not overlap.  Asynchronous on the handle's stream. */
int dcp_inv_f32(dcp_handle* h, const float* X, int64_t batch, int64_t n, float* out);
int dcp_inv_f64(dcp_handle* h, const double* X, int64_t batch, int64_t n, double* out);
int dcp_inv_c64(dcp_handle* h, const void* X, int64_t batch, int64_t n, void* out);
int dcp_inv_c128(dcp_handle* h, const void* X, int64_t batch, int64_t n, void* out);

/* Test hook (not a reference interface): C[M,N] = op(A) . op(B) through the same GEMM
 * cores the solvers use.  form: 0 = NT (A[M,K], B[N,K]), 1 = NN (A[M,K], B[K,N]),
 * 2 = TN (A[K,M], B[K,N]).  ksplits >= 1 selects split-K (partials summed in order).
 * tile: 0 = auto, 1 = large tile, 2 = small tile. */
int dcp_gemm_f32(dcp_handle* h, int form, const float* A, const float* B, float* C,
                 int64_t M, int64_t N, int64_t K, int ksplits, int tile);
int dcp_gemm_f64(dcp_handle* h, int form, const double* A, const double* B, double* C,
                 int64_t M, int64_t N, int64_t K, int ksplits, int tile);
/* complex64: form 0 = A B^H, 1 = A B, 2 = A^H B (the conjugations the solvers use) */
int dcp_gemm_c64(dcp_handle* h, int form, const void* A, const void* B, void* C,
                 int64_t M, int64_t N, int64_t K, int ksplits, int tile);
int dcp_gemm_c128(dcp_handle* h, int form, const void* A, const void* B, void* C,
                  int64_t M, int64_t N, int64_t K, int ksplits, int tile);

/* Test hook (not a reference interface): the reduction-over-samples products (form 2) run the "pair" LDS schedule
 * of the fp32 MFMA core unless DCP_TN_PLAIN is set in the environment (read once); on != 0 selects the plain
 * schedule, on == 0 the pair schedule, on < 0 only queries.  Returns the previous setting (1 = plain).  Process-wide;
 * tests/test_gpu_gemm.py runs the same products both ways so that a toolchain change that breaks the hand-placed
 * LDS waits of the pair schedule is caught. */
int dcp_debug_tn_plain(int on);

/* PMC calibration aid (not a reference interface): reads p[rows, cols] exactly once with the
 * global-load shape of the GEMM panel loaders (pattern 0: 16 rows x 64 B per wave instruction;
 * pattern 1: 512-B row segments), so that rocprofv3 FETCH_SIZE can be compared with a known
 * byte count (tools/calib_fetch.py). */
int dcp_calib_read_f32(dcp_handle* h, const float* p, int64_t rows, int64_t cols, int pattern,
                       float* out);

/* ---- NMF, multiplicative update ------------------------------------------------ */
/* decomp/nmf_methods/batch_mu.py:8-26 (whole loop).  D must already be l2_strict
 * normalised (nmf.py:70).  mask may be NULL.  On return X and D hold what the
 * reference returns: (it, D_new, x) when max|D - D_new| < tol at iteration it,
 * else (maxiter, D, x) after maxiter-1 iterations.  last_maxdiff (host, nullable)
 * receives the last max|D - D_new|; resid_trace (host, nullable, length >= maxiter)
 * receives ||(Y - X D_new) o mask||_F after every iteration (parity metric; costs one
 * extra N.K.F product per iteration, so leave it NULL when timing). */
int dcp_nmf_mu_f32(dcp_handle* h, const float* Y, const float* mask, float* X, float* D,
                   int64_t N, int64_t F, int64_t K, int likelihood, float tol, int maxiter,
                   int* it_out, float* last_maxdiff, float* resid_trace);
int dcp_nmf_mu_f64(dcp_handle* h, const double* Y, const double* mask, double* X, double* D,
                   int64_t N, int64_t F, int64_t K, int likelihood, double tol, int maxiter,
                   int* it_out, double* last_maxdiff, double* resid_trace);

/* The same loop for a problem whose ROWS are sharded over the ranks of the handle's communicator
 * (dcp_comm_init): Y, mask, X are this rank's rows, D is replicated.  Every iteration runs
 *   local x update + [x^T Y | x^T x] (or [num | den]) on this rank's rows
 *   -> ncclAllReduce(sum) of the [K, W] statistics on the handle's stream        (the ONLY exchange)
 *   -> the replicated D update, l2_strict and max|D - D_new| (identical on every rank)
 * with the stop test of iteration i read after iteration i+1 has been enqueued, exactly as in
 * dcp_nmf_mu_*: no host synchronisation and no second stream inside a step.  Every rank must call it
 * with the same F, K, likelihood, tol, maxiter and masked-ness; N may differ per rank.  it_out,
 * last_maxdiff and D are identical on all ranks.  DCP_ERR_COMM without a communicator. */
int dcp_nmf_mu_sharded_f32(dcp_handle* h, const float* Y, const float* mask, float* X, float* D,
                           int64_t N, int64_t F, int64_t K, int likelihood, float tol, int maxiter,
                           int* it_out, float* last_maxdiff);
int dcp_nmf_mu_sharded_f64(dcp_handle* h, const double* Y, const double* mask, double* X, double* D,
                           int64_t N, int64_t F, int64_t K, int likelihood, double tol, int maxiter,
                           int* it_out, double* last_maxdiff);

/* One iteration split at the data-parallel exchange point, asynchronous:
 *   dcp_nmf_mu_stats_*  : X_out <- update_x(X) (grads.py:77-84; X_out may alias X, or be a
 *                         second buffer so that the caller can overlap the stop test of the
 *                         previous iteration), then this rank's share of the D-side sums
 *                         (grads.py:117-125, with the NEW x) into stats:
 *                           l2, no mask : stats[K, F+K] = [ x^T Y | x^T x ]
 *                           otherwise   : stats[K, 2F]  = [ numerator | denominator ]
 *                         (the caller all-reduces stats over ranks: sums over rows)
 *   dcp_nmf_mu_update_* : D_new <- l2_strict(D o max(num,0) / max(den,1e-15))
 *                         (grads.py:86-93, batch_mu.py:21) written to D_new, and
 *                         max|D - D_new| written to the DEVICE scalar maxdiff_dev.
 *                         maxdiff_next (nullable): when given, *maxdiff_dev must be 0 on
 *                         entry, the max is formed by one atomic per row (no extra launch)
 *                         and *maxdiff_next is cleared for the following iteration.
 * stats width: dcp_nmf_mu_stats_width(). */
int64_t dcp_nmf_mu_stats_width(int64_t F, int64_t K, int likelihood, int masked);
int dcp_nmf_mu_stats_f32(dcp_handle* h, const float* Y, const float* mask, const float* X,
                         float* X_out, const float* D, int64_t N, int64_t F, int64_t K,
                         int likelihood, float* stats);
int dcp_nmf_mu_stats_f64(dcp_handle* h, const double* Y, const double* mask, const double* X,
                         double* X_out, const double* D, int64_t N, int64_t F, int64_t K,
                         int likelihood, double* stats);
/* Loop-invariant mask work of a masked run, done ONCE instead of in every dcp_nmf_mu_stats_* call
 * (grads.py:114,124 recompute y * mask per gradient): Ym[N,F] = Y o mask, and -- float32 only, `bits`
 * non-NULL -- the row-bit image of the mask (dcp_nmf_mask_bits_words(N, F) uint32 words: word
 * (row / 32, col) holds mask[32 (row/32) + b, col] != 0 in bit b).  *binary (HOST) = 1 when every mask
 * entry is exactly 0 or 1, i.e. when `bits` may be passed to dcp_nmf_mu_stats_prepared_* (otherwise pass
 * NULL there).  dcp_nmf_mu_stats_prepared_*: dcp_nmf_mu_stats_* on (Ym, mask[, bits]); with bits the
 * (x D) o M products multiply by bits fetched ahead of the GEMM main loop instead of loading the float
 * mask in the epilogue.  Results are identical with and without bits. */
int64_t dcp_nmf_mask_bits_words(int64_t N, int64_t F);
int dcp_nmf_mask_prepare_f32(dcp_handle* h, const float* Y, const float* mask, int64_t N, int64_t F,
                             float* Ym, uint32_t* bits, int* binary);
int dcp_nmf_mask_prepare_f64(dcp_handle* h, const double* Y, const double* mask, int64_t N, int64_t F,
                             double* Ym, uint32_t* bits, int* binary);
int dcp_nmf_mu_stats_prepared_f32(dcp_handle* h, const float* Ym, const float* mask, const uint32_t* bits,
                                  const float* X, float* X_out, const float* D, int64_t N, int64_t F,
                                  int64_t K, int likelihood, float* stats);
int dcp_nmf_mu_stats_prepared_f64(dcp_handle* h, const double* Ym, const double* mask, const uint32_t* bits,
                                  const double* X, double* X_out, const double* D, int64_t N, int64_t F,
                                  int64_t K, int likelihood, double* stats);
int dcp_nmf_mu_update_f32(dcp_handle* h, const float* stats, const float* D, float* D_new,
                          int64_t F, int64_t K, int likelihood, int masked,
                          float* maxdiff_dev, float* maxdiff_next);
int dcp_nmf_mu_update_f64(dcp_handle* h, const double* stats, const double* D, double* D_new,
                          int64_t F, int64_t K, int likelihood, int masked,
                          double* maxdiff_dev, double* maxdiff_next);

/* Building blocks of the stochastic MU variants (decomp/nmf_methods/serizel.py:36-165,
 * kasai.py:36-88), which are host loops over minibatches around the same gradients:
 *   dcp_nmf_grads_*: n_x_updates times  x_mb <- x_mb o max(gx+,0)/max(gx-,1e-15)  in place
 *                    (serizel.py:46-49, kasai.py:63-67), then the two parts of the D gradient
 *                    (grads.py:117-125, 152-160) of that minibatch into grad_pos / grad_neg [K,F].
 *   dcp_nmf_apply_*: D_new = l2_strict(rule) and max|D - D_new| to the host:
 *                    alpha <  0: D o max(P,0)/max(Q,1e-15)                 (serizel.py:54-57)
 *                    alpha >= 0: max(D o ((1-alpha) + alpha P/max(Q,1e-15)), 0)  (kasai.py:77-78)
 *   dcp_axpby_*    : y = a x + b y  (gradient averaging serizel.py:95-96, kasai.py:74-75); a zero
 *                    coefficient means its operand is not read (y may be uninitialised when b = 0);
 *                    x may alias y. */
int dcp_nmf_grads_f32(dcp_handle* h, const float* Y, const float* mask, float* X, const float* D,
                      int64_t N, int64_t F, int64_t K, int likelihood, int n_x_updates,
                      float* grad_pos, float* grad_neg);
int dcp_nmf_grads_f64(dcp_handle* h, const double* Y, const double* mask, double* X, const double* D,
                      int64_t N, int64_t F, int64_t K, int likelihood, int n_x_updates,
                      double* grad_pos, double* grad_neg);
/* Gaussian.grad_x / Poisson.grad_x (decomp/nmf_methods/grads.py:108-115, 143-150), the plugin surface a
 * user subclass reaches through super(): grad_pos, grad_neg [N, K].  Without a mask the l2 negative part
 * is x (D D^T) (the Gram identity of (x D) D^T) and the kl negative part -- the reference's [1, K] row
 * d.T.sum(axis=0) -- is repeated on every row.  X is not modified.  Asynchronous. */
int dcp_nmf_grad_x_f32(dcp_handle* h, const float* Y, const float* mask, const float* X, const float* D,
                       int64_t N, int64_t F, int64_t K, int likelihood, float* grad_pos, float* grad_neg);
int dcp_nmf_grad_x_f64(dcp_handle* h, const double* Y, const double* mask, const double* X, const double* D,
                       int64_t N, int64_t F, int64_t K, int likelihood, double* grad_pos, double* grad_neg);
/* Gaussian.logp (grads.py:127-135): sum((-0.5 ((y - x d) / scale)^2 - log(scale) - pi / 2) [* mask]) to the
 * HOST double (accumulated in double precision).  Synchronises. */
int dcp_nmf_gauss_logp_f32(dcp_handle* h, const float* Y, const float* mask, const float* X, const float* D,
                           int64_t N, int64_t F, int64_t K, double scale, double* out);
int dcp_nmf_gauss_logp_f64(dcp_handle* h, const double* Y, const double* mask, const double* X, const double* D,
                           int64_t N, int64_t F, int64_t K, double scale, double* out);
int dcp_nmf_apply_f32(dcp_handle* h, const float* D, const float* P, const float* Q, double alpha,
                      float* D_new, int64_t K, int64_t F, double* maxdiff);
int dcp_nmf_apply_f64(dcp_handle* h, const double* D, const double* P, const double* Q, double alpha,
                      double* D_new, int64_t K, int64_t F, double* maxdiff);
int dcp_axpby_f32(dcp_handle* h, int64_t n, double a, const float* x, double b, float* y);
int dcp_axpby_f64(dcp_handle* h, int64_t n, double a, const double* x, double b, double* y);

/* Likelihood.update_x / update_d (decomp/nmf_methods/grads.py:77-93), the multiplicative rule a
 * user-supplied Likelihood subclass inherits:  out = cur o max(pos, 0) / max(neg, 1e-15), all
 * [rows, cols] contiguous (out may alias cur).  Asynchronous on the handle's stream. */
int dcp_mu_quotient_f32(dcp_handle* h, const float* cur, const float* pos, const float* neg,
                        int64_t rows, int64_t cols, float* out);
int dcp_mu_quotient_f64(dcp_handle* h, const double* cur, const double* pos, const double* neg,
                        int64_t rows, int64_t cols, double* out);
/* The tail of one MU iteration for a caller-produced U (batch_mu.py:21-22 with a user Likelihood):
 * out = l2_strict(U) (strict != 0) or l2(U), *maxdiff (HOST) = max |ref - out|.  Synchronises. */
int dcp_l2_normalize_diff_f32(dcp_handle* h, const float* U, const float* ref, float* out, int64_t K,
                              int64_t F, int strict, double* maxdiff);
int dcp_l2_normalize_diff_f64(dcp_handle* h, const double* U, const double* ref, double* out, int64_t K,
                              int64_t F, int strict, double* maxdiff);

/* ||(Y - X D) o mask||_F (parity metric of SURVEY 8d; mask nullable). */
int dcp_nmf_residual_f32(dcp_handle* h, const float* Y, const float* mask, const float* X,
                         const float* D, int64_t N, int64_t F, int64_t K, double* out);
int dcp_nmf_residual_f64(dcp_handle* h, const double* Y, const double* mask, const double* X,
                         const double* D, int64_t N, int64_t F, int64_t K, double* out);

/* ---- batched LASSO / NNLS ---------------------------------------------------------- */
/* decomp/lasso.py:97-189 (solve_fastpath, everything after validation):
 *   argmin_x 1/(2n) |y - x A|^2 + alpha |x|_1  for every row of Y[N,F] (batch dims
 *   flattened by the caller), A[K,F], X[N,K] in: initial estimate, out: solution.
 * mask: NULL (mask_ndim 0), [F] (mask_ndim 1, lasso.py:120-122) or [N,F] (mask_ndim 2,
 * lasso.py:160-186), real dtype of the problem.  method: DCP_LASSO_*; positive != 0 selects
 * the `_pos` (NNLS) proximal operator (real dtypes only, lasso.py:92).
 * *it_out (host) is the reference's iteration count: the index of the first iteration
 * i % 10 == 0 at which max(|dx| - tol) < 0, else maxiter - 1 (with the reference's
 * choice of returned iterate per method, lasso.py:297,357,415).  Coordinate descent has no size
 * limit (K <= 2048: a row's coefficients live in one wave's registers; wider: in that row's memory).
 * Synchronises the stream before returning. */
int dcp_lasso_f32(dcp_handle* h, const float* Y, const float* mask, int mask_ndim, const float* A,
                  float* X, int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter,
                  int method, int positive, int* it_out);
int dcp_lasso_f64(dcp_handle* h, const double* Y, const double* mask, int mask_ndim, const double* A,
                  double* X, int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter,
                  int method, int positive, int* it_out);
int dcp_lasso_c64(dcp_handle* h, const void* Y, const float* mask, int mask_ndim, const void* A,
                  void* X, int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter,
                  int method, int positive, int* it_out);
int dcp_lasso_c128(dcp_handle* h, const void* Y, const double* mask, int mask_ndim, const void* A,
                   void* X, int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter,
                   int method, int positive, int* it_out);

/* decomp/lasso.py:448-523 `_solve_parallel_cd(_mask)`: every iteration evaluates the unit-step
 * proximal update of all coordinates and commits it on p = int(K / Gershgorin(A A^H)) of them,
 * chosen by a 0/1 vector that the reference re-shuffles with np.random.RandomState(0) each
 * iteration.  The RNG stream is host state, so the caller supplies it: `order` is a DEVICE
 * int32 [order_rows, K] table, row i = arange(K) after i + 1 cumulative RandomState(0).shuffle
 * calls (a shuffle's swap sequence does not depend on the array's content); coordinate k is
 * committed in iteration i iff order[i, k] < p.  order_rows >= maxiter.  p <= 1: the library
 * runs plain coordinate descent as the reference does (lasso.py:469-470); with a 2-D mask the
 * reference's fallback call is broken (lasso.py:509) and DCP_ERR_REF_TYPEERROR is returned. */
int dcp_lasso_pcd_f32(dcp_handle* h, const float* Y, const float* mask, int mask_ndim, const float* A,
                      float* X, int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter,
                      int positive, const int32_t* order, int64_t order_rows, int* it_out);
int dcp_lasso_pcd_f64(dcp_handle* h, const double* Y, const double* mask, int mask_ndim, const double* A,
                      double* X, int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter,
                      int positive, const int32_t* order, int64_t order_rows, int* it_out);
int dcp_lasso_pcd_c64(dcp_handle* h, const void* Y, const float* mask, int mask_ndim, const void* A,
                      void* X, int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter,
                      int positive, const int32_t* order, int64_t order_rows, int* it_out);
int dcp_lasso_pcd_c128(dcp_handle* h, const void* Y, const double* mask, int mask_ndim, const void* A,
                       void* X, int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter,
                       int positive, const int32_t* order, int64_t order_rows, int* it_out);

/* decomp/lasso.py:586-657 `_solve_admm(_mask)` with penalty rho (dcp_lasso_* with
 * DCP_LASSO_ADMM uses rho = 1.0 like solve_fastpath).  (A A^H + rho I)^-1 is formed on the
 * device in double precision (math_utils/linalg.py:9-16); a 2-D mask needs one K x K system
 * per row (N K^2 workspace, as the reference's lasso.py:643; K <= 10240 real / 5120 complex). */
int dcp_lasso_admm_f32(dcp_handle* h, const float* Y, const float* mask, int mask_ndim, const float* A,
                       float* X, int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter,
                       int positive, double rho, int* it_out);
int dcp_lasso_admm_f64(dcp_handle* h, const double* Y, const double* mask, int mask_ndim, const double* A,
                       double* X, int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter,
                       int positive, double rho, int* it_out);
int dcp_lasso_admm_c64(dcp_handle* h, const void* Y, const float* mask, int mask_ndim, const void* A,
                       void* X, int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter,
                       int positive, double rho, int* it_out);
int dcp_lasso_admm_c128(dcp_handle* h, const void* Y, const double* mask, int mask_ndim, const void* A,
                        void* X, int64_t N, int64_t F, int64_t K, double alpha, double tol, int maxiter,
                        int positive, double rho, int* it_out);

/* ---- row movers of the minibatch containers (decomp/utils/data.py:124-156, 214-313) ------ */
/* gather : out[i, :] = in[index[i], :]      scatter: out[index[i], :] = in[i, :]      i < rows
 * Rows are row_bytes long (any dtype); index: int64 in DEVICE memory.  `in` / `out` may be
 * device memory or PINNED host memory (hipHostMalloc; mapped into the device's address space):
 * the out-of-core container gathers each shuffled minibatch straight out of the host array over
 * PCIe and scatters updated rows back, so no shuffle pass over the host data is ever needed.
 * Asynchronous on the handle's stream. */
int dcp_gather_rows_bytes(dcp_handle* h, const void* in, const int64_t* index, int64_t rows,
                          int64_t row_bytes, void* out);
int dcp_scatter_rows_bytes(dcp_handle* h, const void* in, const int64_t* index, int64_t rows,
                           int64_t row_bytes, void* out);

/* Registers ONE row gather (out[i, :] = in[index[i], :], as dcp_gather_rows_bytes) that the NEXT dcp_dict_step_*
 * / dcp_dict_step_async_* call on this handle runs on the library's side stream beside its atom sweep (after its
 * statistics product, joined before the step's last kernel): the rows of minibatch s + 1 are staged while step s
 * leaves the chip and its HBM nearly idle (decomp/utils/data.py:152-156 gathers all of y once per epoch instead).
 * Use a second staging block for `out`.  Everything enqueued on the
 * handle's stream after that step sees the gathered rows.  rows = 0 clears a pending registration. */
int dcp_dict_prefetch_rows_bytes(dcp_handle* h, const void* in, const int64_t* index, int64_t rows,
                                 int64_t row_bytes, void* out);

/* parallel_cd as the inner solver of the dictionary step (dcp_dict_*): the shuffle table of
 * dcp_lasso_pcd_* (DEVICE int32 [rows, K], rows >= lasso_iter), remembered by the handle until
 * replaced or cleared with order = NULL.  The reference restarts RandomState(0) in every call of the
 * solver (lasso.py:463), so one table serves every minibatch.  The memory stays owned by the caller. */
int dcp_dict_set_pcd_order(dcp_handle* h, const int32_t* order, int64_t rows, int64_t K);

/* ---- online dictionary learning (block coordinate descent) -------------------------- */
/* One minibatch step of decomp/dictionary_learning.py:135-164 (solve_cd), split at the
 * data-parallel exchange point like the NMF step:
 *   dcp_dict_stats_*  : x_mb <- lasso.solve_fastpath(y_mb, D, alpha, x_mb, lasso_tol,
 *                       lasso_iter, lasso_method) (in place), then this rank's
 *                       stats[K, F+K] = x_mb^H [ y_mb | x_mb ]   (lines 137-152)
 *   dcp_dict_update_* : A <- beta A + stats[:, F:], B <- beta B + stats[:, :F] (147-152),
 *                       the sequential atom sweep into D_new (154-159), and max|D - D_new|
 *                       into the DEVICE scalar maxdiff_dev (161).  Asynchronous.
 *   dcp_dict_step_*   : both on one GPU; max|D - D_new| returned to the HOST double.
 *   dcp_dict_step_async_* : the same step with max|D - D_new| left in the DEVICE scalar maxdiff_dev and no
 *                       wait for the GPU: the caller evaluates the stop test (line 161-162) one step late,
 *                       when the next minibatch is already enqueued (decomp_amd.dictionary_learning).
 *                       maxdiff_dev may be device memory or pinned, device-mapped host memory (hipHostMalloc): the
 *                       step's last kernel stores the value with a plain store, so a caller that puts a sentinel
 *                       (-1; max|.| >= 0 or NaN) into a pinned word before the call can poll it instead of
 *                       recording an event -- a recorded event is a barrier packet that idles the stream ~6 us.
 *                       *lasso_it (host) is final when the call returns (a coordinate-descent solve settles it at the
 *                       end of the step, when its stop flag has landed in pinned memory; nothing waits mid-step).
 * D [K,F] must be l2_strict-normalised by the caller on entry of the run (line 126); A [K,K]
 * and B [K,F] are the running statistics (zero before the first step). */
int dcp_dict_stats_f32(dcp_handle* h, const float* Y, float* X, const float* D, int64_t Nb, int64_t F,
                        int64_t K, double alpha, int lasso_method, int lasso_iter, double lasso_tol,
                        float* stats, int* lasso_it);
int dcp_dict_update_f32(dcp_handle* h, const float* stats, double beta, float* A, float* B, const float* D,
                         float* D_new, int64_t F, int64_t K, float* maxdiff_dev);
int dcp_dict_step_f32(dcp_handle* h, const float* Y, float* X, const float* D, float* D_new, float* A, float* B,
                       int64_t Nb, int64_t F, int64_t K, double beta, double alpha, int lasso_method,
                       int lasso_iter, double lasso_tol, double* maxdiff, int* lasso_it);
int dcp_dict_step_async_f32(dcp_handle* h, const float* Y, float* X, const float* D, float* D_new, float* A, float* B,
                             int64_t Nb, int64_t F, int64_t K, double beta, double alpha, int lasso_method,
                             int lasso_iter, double lasso_tol, float* maxdiff_dev, int* lasso_it);
/* out[i, :] = in[index[i], :]  (MinibatchData.shuffle / .array, decomp/utils/data.py:147-156);
 * index: int64 on the device. */
int dcp_gather_rows_f32(dcp_handle* h, const float* in, const int64_t* index, int64_t rows,
                         int64_t cols, float* out);
int dcp_dict_stats_f64(dcp_handle* h, const double* Y, double* X, const double* D, int64_t Nb, int64_t F,
                        int64_t K, double alpha, int lasso_method, int lasso_iter, double lasso_tol,
                        double* stats, int* lasso_it);
int dcp_dict_update_f64(dcp_handle* h, const double* stats, double beta, double* A, double* B, const double* D,
                         double* D_new, int64_t F, int64_t K, double* maxdiff_dev);
int dcp_dict_step_f64(dcp_handle* h, const double* Y, double* X, const double* D, double* D_new, double* A, double* B,
                       int64_t Nb, int64_t F, int64_t K, double beta, double alpha, int lasso_method,
                       int lasso_iter, double lasso_tol, double* maxdiff, int* lasso_it);
int dcp_dict_step_async_f64(dcp_handle* h, const double* Y, double* X, const double* D, double* D_new, double* A, double* B,
                             int64_t Nb, int64_t F, int64_t K, double beta, double alpha, int lasso_method,
                             int lasso_iter, double lasso_tol, double* maxdiff_dev, int* lasso_it);
/* out[i, :] = in[index[i], :]  (MinibatchData.shuffle / .array, decomp/utils/data.py:147-156);
 * index: int64 on the device. */
int dcp_gather_rows_f64(dcp_handle* h, const double* in, const int64_t* index, int64_t rows,
                         int64_t cols, double* out);
int dcp_dict_stats_c64(dcp_handle* h, const void* Y, void* X, const void* D, int64_t Nb, int64_t F,
                        int64_t K, double alpha, int lasso_method, int lasso_iter, double lasso_tol,
                        void* stats, int* lasso_it);
int dcp_dict_update_c64(dcp_handle* h, const void* stats, double beta, void* A, void* B, const void* D,
                         void* D_new, int64_t F, int64_t K, float* maxdiff_dev);
int dcp_dict_step_c64(dcp_handle* h, const void* Y, void* X, const void* D, void* D_new, void* A, void* B,
                       int64_t Nb, int64_t F, int64_t K, double beta, double alpha, int lasso_method,
                       int lasso_iter, double lasso_tol, double* maxdiff, int* lasso_it);
int dcp_dict_step_async_c64(dcp_handle* h, const void* Y, void* X, const void* D, void* D_new, void* A, void* B,
                             int64_t Nb, int64_t F, int64_t K, double beta, double alpha, int lasso_method,
                             int lasso_iter, double lasso_tol, float* maxdiff_dev, int* lasso_it);
/* out[i, :] = in[index[i], :]  (MinibatchData.shuffle / .array, decomp/utils/data.py:147-156);
 * index: int64 on the device. */
int dcp_gather_rows_c64(dcp_handle* h, const void* in, const int64_t* index, int64_t rows,
                         int64_t cols, void* out);
int dcp_dict_stats_c128(dcp_handle* h, const void* Y, void* X, const void* D, int64_t Nb, int64_t F,
                        int64_t K, double alpha, int lasso_method, int lasso_iter, double lasso_tol,
                        void* stats, int* lasso_it);
int dcp_dict_update_c128(dcp_handle* h, const void* stats, double beta, void* A, void* B, const void* D,
                         void* D_new, int64_t F, int64_t K, double* maxdiff_dev);
int dcp_dict_step_c128(dcp_handle* h, const void* Y, void* X, const void* D, void* D_new, void* A, void* B,
                       int64_t Nb, int64_t F, int64_t K, double beta, double alpha, int lasso_method,
                       int lasso_iter, double lasso_tol, double* maxdiff, int* lasso_it);
int dcp_dict_step_async_c128(dcp_handle* h, const void* Y, void* X, const void* D, void* D_new, void* A, void* B,
                             int64_t Nb, int64_t F, int64_t K, double beta, double alpha, int lasso_method,
                             int lasso_iter, double lasso_tol, double* maxdiff_dev, int* lasso_it);
/* out[i, :] = in[index[i], :]  (MinibatchData.shuffle / .array, decomp/utils/data.py:147-156);
 * index: int64 on the device. */
int dcp_gather_rows_c128(dcp_handle* h, const void* in, const int64_t* index, int64_t rows,
                         int64_t cols, void* out);
/* One minibatch step of the MASKED variant, decomp/dictionary_learning.py:192-225
 * (solve_cd_mask): lasso with the 2-D mask, A3[K,F,K] <- beta A3 + x^H (x (x) m) (per-channel
 * Gram, :209-213), B <- beta B + x^H (y o m), the atom update against the OLD dictionary
 * (:218-223) and max|D - D_new| to the host.  A parity path (O(F K^2 Nb) statistic as in the
 * reference), single GPU. */
int dcp_dict_mask_step_f32(dcp_handle* h, const float* Y, const float* mask, float* X, const float* D,
                            float* D_new, float* A3, float* B, int64_t Nb, int64_t F, int64_t K, double beta,
                            double alpha, int lasso_method, int lasso_iter, double lasso_tol,
                            double* maxdiff, int* lasso_it);
int dcp_dict_mask_step_f64(dcp_handle* h, const double* Y, const double* mask, double* X, const double* D,
                            double* D_new, double* A3, double* B, int64_t Nb, int64_t F, int64_t K, double beta,
                            double alpha, int lasso_method, int lasso_iter, double lasso_tol,
                            double* maxdiff, int* lasso_it);
int dcp_dict_mask_step_c64(dcp_handle* h, const void* Y, const float* mask, void* X, const void* D,
                            void* D_new, void* A3, void* B, int64_t Nb, int64_t F, int64_t K, double beta,
                            double alpha, int lasso_method, int lasso_iter, double lasso_tol,
                            double* maxdiff, int* lasso_it);
int dcp_dict_mask_step_c128(dcp_handle* h, const void* Y, const double* mask, void* X, const void* D,
                            void* D_new, void* A3, void* B, int64_t Nb, int64_t F, int64_t K, double beta,
                            double alpha, int lasso_method, int lasso_iter, double lasso_tol,
                            double* maxdiff, int* lasso_it);

#ifdef __cplusplus
}
#endif
#endif /* DECOMP_HIP_H */
