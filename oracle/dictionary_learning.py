"""CPU oracle: online dictionary learning, block coordinate descent
(test infrastructure only).

Restates, in NumPy, the reference's
  decomp/dictionary_learning.py:12-111    entry (x=ones default in D.dtype,
                                          minibatch mandatory, RandomState)
  decomp/dictionary_learning.py:114-168   solve_cd (Mairal et al. block CD)
  decomp/dictionary_learning.py:171-231   solve_cd_mask ([K,F,K] statistics)
with the minibatch containers of decomp/utils/data.py (oracle.common.RowBatches).
"""
import numpy as np
from .common import JITTER, l2, l2_strict, RowBatches, Nones
from . import lasso


def atom_sweep(D, A, B):
    """dictionary_learning.py:154-159 : sequential (Gauss-Seidel) atom update
    u_k = (B_k - A_k . D) / (A_kk + 1e-15) + D_k ;  D_k <- l2(u_k)."""
    D = D.copy()
    for k in range(D.shape[0]):
        u = (B[k] - np.dot(A[k], D)) / (A[k, k] + JITTER) + D[k]
        D[k] = l2(u)
    return D


def atom_sweep_mask(D, D_prev, A, B):
    """dictionary_learning.py:218-223.  QUIRK: the A_k.D contraction uses the
    dictionary from BEFORE the sweep (``D``), not the partially updated one."""
    D_new = D.copy()
    for k in range(D_new.shape[0]):
        AkD = np.einsum('jk,kj->j', A[k], D_prev)
        Akk = np.sum(A[k, :, k] + JITTER)
        u = (B[k] - AkD) / Akk + D_new[k]
        D_new[k] = l2(u)
    return D_new


def minibatch_step(y_mb, x_mb, D, A, B, count, minibatch, alpha, lasso_method,
                   lasso_iter, lasso_tol):
    """One unmasked minibatch step, dictionary_learning.py:137-161: LASSO on the
    minibatch (x_mb is the warm start and is NOT modified here), the running
    statistics A <- beta A + x^H x, B <- beta B + x^H y, the sequential atom
    sweep.  Returns (lasso_it, x_new, A, B, D_new, max|D - D_new|)."""
    it2, x_new = lasso.solve_fastpath(                           # :137-139
        y_mb, D, alpha, x=x_mb, tol=lasso_tol, maxiter=lasso_iter,
        method=lasso_method, mask=None)
    theta = count * minibatch + 1.0                              # :143
    beta = (theta - minibatch) / theta                           # :144 (QUIRK: < 0 at count 0)
    xH = np.conj(x_new.T) if y_mb.dtype.kind == 'c' else x_new.T  # :147-149
    A = beta * A + np.dot(xH, x_new)                             # :151
    B = beta * B + np.dot(xH, y_mb)                              # :152
    D_new = atom_sweep(D, A, B)                                  # :154-159
    diff = float(np.max(np.abs(D - D_new)))                      # :161
    return it2, x_new, A, B, D_new, diff


def solve(y, D, alpha, x=None, tol=1.0e-3, minibatch=None, maxiter=1000,
          lasso_method='cd', lasso_iter=10, lasso_tol=1.0e-5, mask=None,
          random_seed=None, trace=None):
    """dictionary_learning.solve(method='block_cd').  ``trace`` (a list)
    receives, per minibatch step, dict(A, B, D, maxdiff, lasso_it)."""
    if x is None:
        x = np.ones((y.shape[0], D.shape[0]), dtype=D.dtype)     # :58-59
    if minibatch is None:
        raise NotImplementedError('Only online methods are implemented. '
                                  'minibatch is required.')
    rng = np.random.RandomState(random_seed)                     # :85
    yb = RowBatches(y, minibatch)
    xb = RowBatches(x, minibatch)
    mb = Nones() if mask is None else RowBatches(mask, minibatch)
    masked = mask is not None

    K, F = D.shape
    index = np.arange(len(y))                                    # :120
    A = np.zeros((K, F, K) if masked else (K, K), dtype=y.dtype)  # :122 / :179
    B = np.zeros((K, F), dtype=y.dtype)
    D = l2_strict(D)                                             # :126
    conj = y.dtype.kind == 'c'
    count = 0
    for it in range(1, maxiter):                                 # :130
        rng.shuffle(index)          # cumulative: the index array itself keeps
        yb.shuffle(index)           # being reshuffled and re-applied (:131-133)
        xb.shuffle(index)
        mb.shuffle(index)
        for y_mb, x_mb, m_mb in zip(yb, xb, mb):
            it2, x_new = lasso.solve_fastpath(                   # :137-139
                y_mb, D, alpha, x=x_mb, tol=lasso_tol, maxiter=lasso_iter,
                method=lasso_method, mask=m_mb)
            x_mb[...] = x_new                                    # :140
            theta = count * minibatch + 1.0                      # :143
            beta = (theta - minibatch) / theta                   # :144 (QUIRK: < 0 at count 0)
            xH = np.conj(x_mb.T) if conj else x_mb.T             # :147-149
            if not masked:
                A = beta * A + np.dot(xH, x_mb)                  # :151
                B = beta * B + np.dot(xH, y_mb)                  # :152
                D_new = atom_sweep(D, A, B)
            else:
                A = beta * A + np.tensordot(                     # :209-213
                    xH, np.expand_dims(x_mb, -2) * np.expand_dims(m_mb, -1),
                    axes=1)
                B = beta * B + np.dot(xH, y_mb * m_mb)           # :214
                D_new = atom_sweep_mask(D, D, A, B)
            diff = float(np.max(np.abs(D - D_new)))
            if trace is not None:
                trace.append({'A': A.copy(), 'B': B.copy(), 'D': D_new.copy(),
                              'maxdiff': diff, 'lasso_it': it2})
            if diff < tol:                                       # :161-162
                return it, D_new, xb.array
            D = D_new
            count += 1
    return maxiter, D, xb.array
