"""dtype helpers (reference: decomp/utils/dtype.py:5-14)."""
import numpy as np

from .exceptions import DtypeMismatchError


def float_type(dtype):
    """The real dtype that carries the magnitude of ``dtype``
    (complex64 -> float32, complex128 -> float64, floats unchanged)."""
    dtype = np.dtype(dtype)
    if dtype.kind == 'f':
        return dtype
    if dtype == np.complex64:
        return np.float32
    if dtype == np.complex128:
        return np.float64
    raise DtypeMismatchError('Invalid dtype is given: ' + str(dtype))
