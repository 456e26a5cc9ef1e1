"""Non-negative least squares with an l1 penalty -- drop-in for ``decomp.nnls``.

The reference's wrapper (decomp/nnls.py:4-7) appends ``'_pos'`` to the method name and
calls ``lasso.solve``.  QUIRK kept: its default ``method='ista_pos'`` therefore becomes
``'ista_pos_pos'``, which ``lasso.solve`` rejects with ValueError -- pass
``method='ista'`` (or any name of ``lasso.AVAILABLE_METHODS``) explicitly.
"""
from . import lasso


def solve(y, A, alpha, x=None, tol=1.0e-3, method='ista_pos', maxiter=1000,
          mask=None, **kwargs):
    return lasso.solve(y, A, alpha, x, tol, method + '_pos', maxiter, mask, **kwargs)
