"""dtype helpers -- the counterpart of the reference's decomp/utils/dtype.py:5-14.

``float_type`` answers "which real dtype carries the magnitudes of this dtype": masks,
norms, step sizes and tolerances of a complex problem live in that dtype (lasso.py:473,
dictionary_learning.py:133).  Table driven here; same answers and the same error type as
the reference for anything that is neither floating nor complex64 / complex128.
"""
import numpy as np

from .exceptions import DtypeMismatchError

_REAL_OF_COMPLEX = {
    np.dtype(np.complex64): np.float32,
    np.dtype(np.complex128): np.float64,
}


def float_type(dtype):
    dt = np.dtype(dtype)
    if dt.kind == 'f':          # float16 / float32 / float64 stay what they are
        return dt
    real = _REAL_OF_COMPLEX.get(dt)
    if real is None:
        raise DtypeMismatchError('Invalid dtype is given: ' + str(dt))
    return real
