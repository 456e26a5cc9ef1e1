"""CPU oracle: stochastic multiplicative-update NMF (test infrastructure only).

Restates, in NumPy, the reference's
  decomp/nmf_methods/serizel.py:36-165  asg / gsg / asag / gsag-MU (Serizel et al.)
  decomp/nmf_methods/kasai.py:36-88     SVRMU / SVRMU-ACC (Kasai)
on the minibatch containers of decomp/utils/data.py (oracle.common.RowBatches).
"""
import numpy as np

from .common import JITTER, l2_strict, RowBatches, Nones
from .nmf import _parts_x, _parts_d, _quotient


def _containers(y, x, mask, minibatch):
    yb, xb = RowBatches(y, minibatch), RowBatches(x, minibatch)
    mb = Nones() if mask is None else RowBatches(mask, minibatch)
    return yb, xb, mb


def _x_step(y_mb, x_mb, D, m_mb, lik):
    x_mb[:] = _quotient(x_mb, *_parts_x(y_mb, x_mb, D, m_mb, lik))


def solve_serizel(y, D, x, tol, minibatch, maxiter, method, likelihood, mask, random_seed,
                  forget_rate=0.5):
    """serizel.py:9-165.  QUIRKS: 'gsg-mu' runs the asg algorithm (:23-25); on convergence
    the old D is returned (:58-59, :122-123, :160-161)."""
    rng = np.random.RandomState(random_seed)
    D = l2_strict(D)                                   # nmf.py:70
    yb, xb, mb = _containers(y, x, mask, minibatch)
    averaged = method in ('asag-mu', 'gsag-mu')
    per_mb = method in ('asg-mu', 'gsg-mu', 'asag-mu')
    index = np.arange(len(y))
    for it in range(1, maxiter):
        rng.shuffle(index)
        yb.shuffle(index); xb.shuffle(index); mb.shuffle(index)
        if averaged:
            spos, sneg = np.zeros_like(D), np.zeros_like(D)
        for y_mb, x_mb, m_mb in zip(yb, xb, mb):
            _x_step(y_mb, x_mb, D, m_mb, likelihood)
            gpos, gneg = _parts_d(y_mb, x_mb, D, m_mb, likelihood)
            P, Q = gpos, gneg
            if averaged:                               # :95-96
                spos = (1.0 - forget_rate) * spos + forget_rate * gpos
                sneg = (1.0 - forget_rate) * sneg + forget_rate * gneg
                P, Q = spos, sneg
            if per_mb:
                D_new = l2_strict(_quotient(D, P, Q))
                if np.max(np.abs(D - D_new)) < tol:
                    return it, D, xb.array
                D = D_new
        if not per_mb:
            D_new = l2_strict(_quotient(D, spos, sneg))
            if np.max(np.abs(D - D_new)) < tol:
                return it, D, xb.array
            D = D_new
    return maxiter, D, xb.array


def solve_kasai(y, D, x, tol, minibatch, maxiter, method, likelihood, mask, random_seed,
                alpha=1.0, beta=0.5):
    """kasai.py:10-88."""
    rng = np.random.RandomState(random_seed)
    D = l2_strict(D)
    if method == 'svrmu':
        iters = 1
    else:                                              # :24-28 (F, K = D.shape as written)
        F, K = D.shape
        N = x.shape[0]
        iters = int(np.maximum(beta * F * (3 * K + 2 * N) / (3 * F * N + 2 * K), 1.0))
    yb, xb, mb = _containers(y, x, mask, minibatch)
    index = np.arange(len(y))
    rng.shuffle(index)                                 # shuffled once (:42-46)
    yb.shuffle(index); xb.shuffle(index); mb.shuffle(index)
    n_mb = yb.n_loop
    prev_pos = np.zeros((n_mb,) + D.shape, dtype=D.dtype)
    prev_neg = np.zeros((n_mb,) + D.shape, dtype=D.dtype)
    for it in range(1, maxiter):
        full_pos, full_neg = np.zeros_like(D), np.zeros_like(D)
        for y_mb, x_mb, m_mb in zip(yb, xb, mb):
            gp, gn = _parts_d(y_mb, x_mb, D, m_mb, likelihood)
            full_pos += gp
            full_neg += gn
        full_pos /= n_mb
        full_neg /= n_mb
        for k, (y_mb, x_mb, m_mb) in enumerate(zip(yb, xb, mb)):
            for _ in range(iters):
                _x_step(y_mb, x_mb, D, m_mb, likelihood)
            gp, gn = _parts_d(y_mb, x_mb, D, m_mb, likelihood)
            P = gp + prev_neg[k] + full_pos            # :74-75
            Q = gn + prev_pos[k] + full_neg
            D_new = D * ((1.0 - alpha) + alpha * P / np.maximum(Q, JITTER))
            D_new = l2_strict(np.maximum(D_new, 0.0))
            if np.max(np.abs(D - D_new)) < tol:
                return it, D, xb.array
            D = D_new
            prev_pos[k] = gp
            prev_neg[k] = gn
    return maxiter, D, xb.array


def solve(y, D, x=None, tol=1.0e-3, minibatch=None, maxiter=1000, method='asg-mu',
          likelihood='l2', mask=None, random_seed=None, **kwargs):
    if x is None:
        x = np.ones((y.shape[0], D.shape[0]), dtype=y.dtype)
    if method in ('asg-mu', 'gsg-mu', 'asag-mu', 'gsag-mu'):
        return solve_serizel(y, D, x, tol, minibatch, maxiter, method, likelihood, mask,
                             random_seed, **kwargs)
    return solve_kasai(y, D, x, tol, minibatch, maxiter, method, likelihood, mask, random_seed,
                       **kwargs)
