"""CPU oracle: batched LASSO / NNLS inner solves (test infrastructure only).

Restates, in NumPy, the reference's
  decomp/lasso.py:97-189    solve_fastpath (row-normalise A, rescale alpha/tol/x)
  decomp/lasso.py:192-241   soft-threshold operators (real / complex / positive)
  decomp/lasso.py:244-271   one proximal-gradient step (plain and masked)
  decomp/lasso.py:274-445   ista / acc_ista / fista, plain and 2-D-masked
  decomp/lasso.py:526-583   coordinate descent, plain and masked (as written)
  decomp/lasso.py:448-523   parallel (shotgun) coordinate descent, plain and masked
  decomp/lasso.py:586-657   ADMM, plain and masked (+ math_utils/linalg.py:9-16 inverse)
The three proximal-gradient solvers share one loop here (they differ only in
the extrapolation coefficient and in which iterate is returned on exhaustion);
the reference's quirks are kept and marked QUIRK.

dtype note: the reference's fista multiplies by an ``np.float64`` scalar, which
under NumPy >= 2 promotes a float32 problem to float64 after the first
iteration (lasso.py:411-412).  This restatement keeps the input dtype; the two
agree to float32 rounding, which is the tolerance the fixtures are checked at.
"""
import numpy as np
from .common import JITTER, gershgorin, real_dtype

METHODS = ('ista', 'acc_ista', 'fista', 'cd', 'parallel_cd', 'admm')


# ----------------------------------------------------------------- prox ----
def shrink_real(z, t):
    """lasso.py:192-207"""
    return np.maximum(np.abs(z) - t, 0.0) * np.sign(z)


def shrink_complex(z, t):
    """lasso.py:210-225 : shrink the modulus, keep the phase (jitter in the
    phase denominator)."""
    r = np.abs(z)
    return np.maximum(r - t, 0.0) * (z / (r + JITTER))


def shrink_positive(z, t):
    """lasso.py:228-241"""
    return np.maximum(z - t, 0.0)


def _pick_shrink(A, positive):
    if positive:
        return shrink_positive
    return shrink_complex if A.dtype.kind == 'c' else shrink_real


def _adjoint(A, positive):
    """lasso.py:276-284 : A^T for real / positive, conj(A^T) for complex."""
    if positive or A.dtype.kind != 'c':
        return A.T
    return np.conj(A.T)


def _mean_over_batch(m):
    """lasso.py:300-303"""
    while m.ndim > 1:
        m = np.mean(m, 0)
    return m


# ---------------------------------------------- proximal-gradient family ----
def _prox_grad(y, A, alpha, x0, tol, maxiter, positive, mask, momentum):
    """ista / acc_ista / fista with or without a full (batch-shaped) mask.

    lasso.py:274-297, 306-328 (ista) ; 331-357, 360-385 (acc_ista) ;
    388-415, 418-445 (fista).
    """
    rdt = real_dtype(y.dtype)
    shrink = _pick_shrink(A, positive)
    At = _adjoint(A, positive)
    if mask is None:
        AAt = A.dot(At)                                        # :285
        yAt = np.tensordot(y, At, axes=1)                      # :289
    else:
        AAt = (A * _mean_over_batch(mask)).dot(At)             # :317
        yAt = np.tensordot(y * mask, At, axes=1)               # :321
    L_inv = 1.0 / gershgorin(AAt)                              # :286  shape (1,)
    thr = L_inv * alpha                                        # :287

    def step(v):                                               # :244-271
        if mask is None:
            back = np.tensordot(v, AAt, axes=1)
        else:
            back = np.tensordot(np.tensordot(v, A, axes=1) * mask, At, axes=1)
        return shrink(v + L_inv * (yAt - back), thr)

    x_prev = x0          # iterate the stop test compares against
    x_new = x0
    v = x0               # extrapolated point fed to the step
    beta = 1.0
    for i in range(maxiter):
        if momentum == 'acc_ista':
            x_prev = x_new                                     # :351
        x_new = step(v)
        if momentum == 'acc_ista':
            c = rdt.type(i / (i + 3))
            v = x_new + c * (x_new - x_prev)                   # :353
        if i % 10 == 0 and np.max(np.abs(x_new - x_prev) - tol) < 0.0:
            return i, x_new                                    # :293-294
        if momentum == 'ista':
            x_prev = x_new
            v = x_new
        elif momentum == 'fista':
            beta_new = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * beta * beta))
            c = rdt.type((beta - 1.0) / beta_new)
            v = x_new + c * (x_new - x_prev)                   # :412
            x_prev = x_new
            beta = beta_new
    # QUIRK: on exhaustion ista/fista hand back the latest iterate, acc_ista
    # the one before it (its `x0 = x0_new` sits at the top of the loop body).
    return maxiter - 1, x_prev


# ------------------------------------------------- coordinate descent ------
def _cd(y, A, alpha, x, tol, maxiter, positive, mask):
    """lasso.py:526-552 and 555-583, as written: x.A is recomputed for every
    coordinate, x is updated in place, the stop test is collected over all
    coordinates on sweeps 0, 10, 20, ..."""
    shrink = _pick_shrink(A, positive)
    At = _adjoint(A, positive)
    if mask is not None:
        y = y * mask
    K = x.shape[-1]
    for i in range(maxiter):
        ok = True
        for k in range(K):
            xA = np.tensordot(x, A, axes=1)
            if mask is not None:
                xA = xA * mask       # QUIRK (:570-572): the x_k A_k term below
            xA = xA - np.tensordot(x[..., k:k + 1], A[k:k + 1], axes=1)  # is unmasked
            z = np.tensordot(y - xA, At[:, k], axes=1)
            z = shrink(z, alpha[..., k])
            if i % 10 == 0:
                ok = ok and bool(np.max(np.abs(x[..., k] - z) - tol[k]) < 0.0)
            x[..., k] = z
        if i % 10 == 0 and ok:
            return i, x
    return maxiter - 1, x


# ------------------------------------------- parallel coordinate descent ---
def _parallel_cd(y, A, alpha, x0, tol, maxiter, positive, mask):
    """lasso.py:448-484 (plain) and 487-523 (2-D mask).

    Every iteration evaluates the full unit-step proximal update, then commits it
    only on the p coordinates selected by a 0/1 vector that is re-shuffled by
    RandomState(0) each iteration (cumulative shuffles of ONE vector).  p =
    int(K / Gershgorin(AAt)); AAt is the unmasked Gram matrix in both variants.
    """
    shrink = _pick_shrink(A, positive)
    At = _adjoint(A, positive)
    rng = np.random.RandomState(0)                             # :463
    AAt = A.dot(At)                                            # :464 / :503
    rho = gershgorin(AAt)
    K = A.shape[0]
    p = int((K / rho).reshape(-1)[0])                          # :468
    if p <= 1:
        if mask is not None:
            # QUIRK (:509): the masked fallback call drops `positive`, so the
            # reference dies with a TypeError (missing argument 'xp').
            raise TypeError("_solve_cd_mask() missing 1 required positional "
                            "argument: 'xp'")
        return _cd(y, A, alpha, x0, tol, maxiter, positive, None)
    if mask is None:
        yAt = np.tensordot(y, At, axes=1)                      # :472
    else:
        yAt = np.tensordot(y * mask, At, axes=1)               # :511
    select = np.zeros(K, dtype=real_dtype(y.dtype))
    select[:p] = 1.0                                           # :473-474
    for i in range(maxiter):
        if mask is None:
            back = np.tensordot(x0, AAt, axes=1)
        else:
            back = np.tensordot(np.tensordot(x0, A, axes=1) * mask, At, axes=1)
        x_new = shrink(x0 + 1.0 * (yAt - back), alpha)         # step 1, threshold alpha
        dx = x_new - x0
        if i % 10 == 0 and np.max(np.abs(dx) - tol) < 0.0:
            return i, x_new
        rng.shuffle(select)                                    # :481
        x0 += dx * select                                      # in place
    return maxiter - 1, x0


# ----------------------------------------------------------------- ADMM ----
def _admm(y, A, alpha, x, tol, maxiter, positive, mask, rho=1.0):
    """lasso.py:586-618 (plain) and 621-657 (2-D mask: one K x K system per row).

    QUIRK: `AAt + rho * eye(K)` adds a float64 identity, so a float32 / complex64
    problem is silently promoted and iterated in double precision; the result
    comes back as float64 / complex128.
    """
    shrink = _pick_shrink(A, positive)
    At = _adjoint(A, positive)
    K = A.shape[0]
    if mask is None:
        yAt = np.tensordot(y, At, axes=1)                      # :601
        AAt = A.dot(At)
        system_inv = np.linalg.inv(AAt + rho * np.eye(K))      # :603

        def apply_inv(v):
            return v.dot(system_inv)
        squeeze = False
    else:
        yAt = np.tensordot(y * mask, At, axes=1)[..., None, :]  # :638
        x = x[..., None, :]
        alpha = alpha[..., None, :]
        tol = tol[None, :]
        AAt = np.tensordot(mask[..., None, :] * A, At, axes=1)  # :643  [..., K, K]
        system_inv = np.linalg.inv(AAt + rho * np.eye(K))

        def apply_inv(v):
            return np.matmul(v, system_inv)
        squeeze = True
    thr = alpha / rho
    u = x.copy()
    z = x.copy()
    for i in range(maxiter):
        x_new = apply_inv(yAt + rho * (z - u))                 # :609
        z = shrink(x_new + u, thr)
        if i % 10 == 0 and (np.max(np.abs(x - x_new) - tol) < 0.0 and
                            np.max(np.abs(z - x_new) - tol) < 0.0):
            return i, (x_new[..., 0, :] if squeeze else x_new)
        x = x_new
        u = u + x - z
    return maxiter - 1, (x[..., 0, :] if squeeze else x)


# --------------------------------------------------------- fast path -------
def solve_fastpath(y, A, alpha, x, tol, maxiter, method, mask=None):
    """lasso.py:97-189."""
    positive = method.endswith('_pos')
    if positive:
        method = method[:-4]
    if method not in METHODS:
        raise NotImplementedError(method)

    if mask is not None and mask.ndim == 1:                    # :120-122
        y = y * mask
        A = A * mask
    if A.dtype.kind == 'c':                                    # :124-127
        s = np.sqrt(np.sum(np.real(np.conj(A) * A), axis=-1))
    else:
        s = np.sqrt(np.sum(np.square(A), axis=-1))
    A = A / s[:, None]                                         # :128
    alpha = alpha / s                                          # :129
    tol = tol * s                                              # :130
    x = x * s                                                  # :131

    if mask is None or mask.ndim == 1:
        n_valid = A.shape[-1] if mask is None else np.sum(mask, axis=-1)
        alpha = alpha * n_valid                                # :135-138
        full_mask = None
    else:
        alpha = alpha * np.sum(mask, axis=-1, keepdims=True)   # :163
        full_mask = mask

    if method == 'cd':
        it, x = _cd(y, A, alpha, x, tol, maxiter, positive, full_mask)
    elif method == 'parallel_cd':
        it, x = _parallel_cd(y, A, alpha, x, tol, maxiter, positive, full_mask)
    elif method == 'admm':
        it, x = _admm(y, A, alpha, x, tol, maxiter, positive, full_mask)
    else:
        it, x = _prox_grad(y, A, alpha, x, tol, maxiter, positive, full_mask,
                           method)
    return it, x / s                                           # :189


def solve(y, A, alpha, x=None, tol=1.0e-3, method='ista', maxiter=1000,
          mask=None):
    """lasso.py:19-94 without the validation (that is host logic of the
    product): default x = zeros(y.shape[:-1] + (K,))."""
    if x is None:
        x = np.zeros(y.shape[:-1] + (A.shape[0],), dtype=y.dtype)
    return solve_fastpath(y, A, alpha, x, tol, maxiter, method, mask=mask)
