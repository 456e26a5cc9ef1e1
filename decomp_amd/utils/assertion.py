"""Argument validation of the drop-in API: pure host logic, runs before anything is
sent to the GPU.  Mirrors the checks (and the exception types) of the reference's
decomp/utils/assertion.py:5-100; works on NumPy arrays and torch tensors alike.
"""
import numpy as np

from .exceptions import ShapeMismatchError, DtypeMismatchError, DimInvalidError
from .. import _arrays


def assert_shapes(x_name, x, y_name, y, axes=None):
    """axes None: identical shapes.  axes = n (int): x.shape[-n:] == y.shape[:n].
    axes = list: the listed axes agree.  None arrays are skipped.
    (assertion.py:5-42)"""
    if x is None or y is None:
        return
    xs, ys = tuple(x.shape), tuple(y.shape)
    if axes is None:
        if xs != ys:
            raise ShapeMismatchError('%s%s and %s%s must have the same shape'
                                     % (x_name, xs, y_name, ys))
        return
    if isinstance(axes, int):
        if xs[-axes:] != ys[:axes]:
            raise ShapeMismatchError(
                'trailing %d axes of %s%s must equal leading %d axes of %s%s'
                % (axes, x_name, xs, axes, y_name, ys))
        return
    if isinstance(axes, (list, tuple)):
        try:
            bad = any(xs[a] != ys[a] for a in axes)
        except IndexError:
            bad = True
        if bad:
            raise ShapeMismatchError('%s%s and %s%s must agree on axes %s'
                                     % (x_name, xs, y_name, ys, list(axes)))
        return
    raise TypeError('Argument axes is invalid, given ' + str(axes))


def assert_ndim(x_name, x, ndim):
    """assertion.py:45-51"""
    if x is None:
        return
    if len(x.shape) != ndim:
        raise DimInvalidError('%s must have %d dimensions, has %d'
                              % (x_name, ndim, len(x.shape)))


def assert_dtypes(dtypes='fc', **arrays):
    """All non-None arrays share one dtype, whose kind is in ``dtypes``
    ('f' float, 'c' complex).  (assertion.py:54-84)"""
    first = None
    for name, a in arrays.items():
        if a is None:
            continue
        dt = _arrays.np_dtype(a)
        if first is None:
            first = (name, dt)
        if dt != first[1]:
            raise DtypeMismatchError('%s is %s but %s is %s: dtypes must be identical'
                                     % (first[0], first[1], name, dt))
        if dt.kind not in dtypes:
            raise DtypeMismatchError('%s has dtype %s, allowed kinds: %s'
                                     % (name, dt, dtypes))


def assert_nonnegative(x):
    """assertion.py:95-100: every element satisfies x >= 0 (NaN fails).  ``x`` is a
    device array (see _arrays.DeviceArrays); the scan runs in a HIP kernel."""
    if x is None:
        return
    assert _arrays.np_dtype(x).kind != 'c'
    assert _arrays.count_negative(x) == 0


def assert_nonnegative_host_or_device(x):
    """assert_nonnegative for an array that has not been sent to the GPU yet (the LASSO
    mask, lasso.py:78): NumPy arrays are scanned where they are, torch CUDA tensors by the
    HIP scan kernel.  Host-side argument validation, not part of the compute path."""
    if x is None:
        return
    if _arrays.is_torch(x):
        assert_nonnegative(x.contiguous())
        return
    assert x.dtype.kind != 'c'
    assert bool((x >= 0.0).all())
