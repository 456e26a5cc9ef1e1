"""Non-negative least squares with an l1 penalty on MI355X -- the ``decomp.nnls`` entry point.

In the reference this module is a three-line forwarder (decomp/nnls.py:4-7): the solver name
gets a ``'_pos'`` suffix and everything else goes to ``lasso.solve``, whose ``*_pos`` methods
use the non-negative proximal operator ``max(z - t, 0)`` (decomp/lasso.py:228-241).  Here
those methods run in libdecomp_hip.so (``positive=1`` of ``dcp_lasso_*``, see
include/decomp_hip.h); this file only reproduces the forwarding, including its quirk:

QUIRK kept on purpose: the default ``method='ista_pos'`` becomes ``'ista_pos_pos'``, which
``lasso.solve`` rejects with ``ValueError`` -- callers have to name a base solver
(``'ista'``, ``'fista'``, ``'cd'``, ...), exactly as with the reference.
"""
from . import lasso as _lasso

_SUFFIX = '_pos'


def solve(y, A, alpha, x=None, tol=1.0e-3, method='ista_pos', maxiter=1000, mask=None, **kwargs):
    """argmin_{x >= 0} 1/(2n) |y - x A|^2 + alpha |x|_1 for every row of ``y``.

    Arguments and the returned ``(it, x)`` are those of ``lasso.solve``; ``method`` is the
    name of a base solver from ``lasso.AVAILABLE_METHODS``.
    """
    positive_method = method + _SUFFIX
    return _lasso.solve(y=y, A=A, alpha=alpha, x=x, tol=tol, method=positive_method,
                        maxiter=maxiter, mask=mask, **kwargs)
