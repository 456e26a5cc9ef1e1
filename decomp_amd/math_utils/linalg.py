"""``decomp.math_utils.linalg`` (linalg.py:9-38): batched inverse.

The only hot-path user is ADMM's (AA^H + rho I)^-1, which runs inside ``dcp_lasso_admm_*``
(Gauss-Jordan kernels, csrc/lasso_extra.hpp); a general-purpose ``inv`` is not exported by
the library."""


def inv(x):
    raise NotImplementedError('linalg.inv is internal to dcp_lasso_admm_* on MI355X '
                              '(csrc/lasso_extra.hpp); it is not a public entry point')
