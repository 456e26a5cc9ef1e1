"""CPU oracle: NMF multiplicative update (test infrastructure only).

Restates, in NumPy, the full-batch 'mu' path of the reference:
  decomp/nmf.py:16-80                     entry: x=ones default, l2_strict(D)
  decomp/nmf_methods/batch_mu.py:8-26     outer loop + stop rule
  decomp/nmf_methods/grads.py:77-93       multiplicative quotient
  decomp/nmf_methods/grads.py:108-125     Gaussian (l2) gradient parts
  decomp/nmf_methods/grads.py:143-160     Poisson (kl) gradient parts
The GEMM structure (f = x.D materialised, six products per iteration) is kept
as in the reference so that fp64 results agree to rounding.
"""
import numpy as np
from .common import JITTER, l2_strict


def _parts_x(y, x, d, mask, likelihood):
    """(positive, negative) parts of the x-gradient."""
    if likelihood == 'l2':            # grads.py:108-115
        f = x.dot(d)
        if mask is not None:
            f = f * mask
            y = y * mask
        return y.dot(d.T), f.dot(d.T)
    # 'kl'                            # grads.py:143-150
    f = x.dot(d) + JITTER
    if mask is None:
        return (y / f).dot(d.T), d.T.sum(axis=0, keepdims=True)
    return ((y * mask) / f).dot(d.T), mask.dot(d.T)


def _parts_d(y, x, d, mask, likelihood):
    """(positive, negative) parts of the D-gradient."""
    if likelihood == 'l2':            # grads.py:117-125
        f = x.dot(d)
        if mask is not None:
            f = f * mask
            y = y * mask
        return x.T.dot(y), x.T.dot(f)
    # 'kl'                            # grads.py:152-160
    f = x.dot(d) + JITTER
    if mask is None:
        return x.T.dot(y / f), x.T.sum(axis=1, keepdims=True)
    return x.T.dot((y * mask) / f), x.T.dot(mask)


def _quotient(cur, pos, neg):
    """grads.py:84,93 : cur * max(pos, 0) / max(neg, 1e-15)."""
    return cur * np.maximum(pos, 0.0) / np.maximum(neg, JITTER)


def update_x(y, x, d, mask=None, likelihood='l2'):
    return _quotient(x, *_parts_x(y, x, d, mask, likelihood))


def update_d(y, x, d, mask=None, likelihood='l2'):
    return _quotient(d, *_parts_d(y, x, d, mask, likelihood))


def residual(y, x, d, mask=None):
    """|| (y - x d) o mask ||_F  (not in the reference; parity metric of
    SURVEY 8d)."""
    r = y - x.dot(d)
    if mask is not None:
        r = r * mask
    return float(np.sqrt(np.sum(r.astype(np.float64) ** 2)))


def mu_step(y, x, d, mask=None, likelihood='l2'):
    """One iteration of batch_mu.py:16-24 -> (x_new, D_new normalised,
    max|D - D_new|)."""
    x = update_x(y, x, d, mask, likelihood)
    d_new = l2_strict(update_d(y, x, d, mask, likelihood))
    return x, d_new, float(np.max(np.abs(d - d_new)))


def mu_step_gram(y, x, d):
    """The product's algorithmic formulation of the unmasked l2 step
    (SURVEY 2.2 k2/k5): x.(D D^T) and (x^T x).D instead of (x D) D^T and
    x^T (x D).  Used only to quantify the rounding drift of that choice."""
    g = d.dot(d.T)
    x = _quotient(x, y.dot(d.T), x.dot(g))
    s = x.T.dot(x)
    d_new = l2_strict(_quotient(d, x.T.dot(y), s.dot(d)))
    return x, d_new, float(np.max(np.abs(d - d_new)))


def solve(y, D, x=None, tol=1.0e-3, maxiter=1000, likelihood='l2', mask=None,
          trace=None):
    """nmf.py:52-78 + batch_mu.py:8-26 (validation lives in the product's host
    layer, not here).  ``trace`` (a list) receives one dict per iteration with
    'maxdiff' and 'resid' (residual after the iteration's x and normalised D).
    Returns (it, D, x) with the reference's conventions: ``range(1, maxiter)``
    iterations, (it, D_new, x) on convergence, (maxiter, D, x) on exhaustion.
    """
    if likelihood in ('gaussian',):
        likelihood = 'l2'
    if likelihood in ('poisson',):
        likelihood = 'kl'
    if x is None:
        x = np.ones((y.shape[0], D.shape[0]), dtype=y.dtype)   # nmf.py:53-54
    D = l2_strict(D)                                           # nmf.py:70
    for it in range(1, maxiter):                               # batch_mu.py:16
        x, d_new, diff = mu_step(y, x, D, mask, likelihood)
        if trace is not None:
            trace.append({'maxdiff': diff,
                          'resid': residual(y, x, d_new, mask)})
        if diff < tol:                                         # batch_mu.py:22
            return it, d_new, x
        D = d_new
    return maxiter, D, x
