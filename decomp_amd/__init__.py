"""decomp_amd -- the iterative-update hot path of deComP (NMF multiplicative update,
batched LASSO inner solves, online dictionary learning) on AMD MI355X.

Drop-in for the reference's ``decomp.nmf.solve`` / ``decomp.lasso.solve`` /
``decomp.dictionary_learning.solve``:

    import decomp_amd as decomp
    it, D, x = decomp.nmf.solve(y, D0, tol=1e-4, maxiter=1000)

All arithmetic runs in hand-written HIP kernels (libdecomp_hip.so, gfx950) behind a
plain C ABI (include/decomp_hip.h); there is no CPU fallback.
"""
from . import nmf, lasso, nnls, dictionary_learning  # noqa: F401
from . import utils, math_utils, nmf_methods  # noqa: F401
from .utils import exceptions  # noqa: F401

__version__ = '0.1.0'
