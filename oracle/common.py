"""Shared helpers of the CPU oracle (test infrastructure only).

Reference sources restated here:
  decomp/utils/normalize.py:2-21      (l2 / l2_strict row normalisation)
  decomp/math_utils/eigen.py:9-20     (Gershgorin bound)
  decomp/utils/dtype.py:5-14          (complex -> real dtype)
  decomp/utils/data.py:63-156,316-330 (minibatch containers)
"""
import numpy as np

JITTER = 1.0e-15


def real_dtype(dt):
    """decomp/utils/dtype.py:5-14"""
    dt = np.dtype(dt)
    if dt.kind == 'f':
        return dt
    if dt == np.complex64:
        return np.dtype(np.float32)
    if dt == np.complex128:
        return np.dtype(np.float64)
    raise ValueError('not a float/complex dtype: ' + str(dt))


def row_sq_norm(U):
    """sum_j |U_ij|^2 with keepdims (normalize.py:6-9 / 17-20)."""
    if U.dtype.kind == 'c':
        return np.sum(np.real(np.conj(U) * U), axis=-1, keepdims=True)
    return np.sum(U * U, axis=-1, keepdims=True)


def l2(U):
    """normalize.py:2-10 : U / sqrt(max(|U|^2, 1))."""
    return U / np.sqrt(np.maximum(row_sq_norm(U), 1.0))


def l2_strict(U):
    """normalize.py:13-21 : U / |U|."""
    return U / np.sqrt(row_sq_norm(U))


def gershgorin(X):
    """eigen.py:20 : max_j sum_i |X_ij|, shape [..., 1]."""
    return np.max(np.sum(np.abs(X), axis=-2), axis=-1, keepdims=True)


class RowBatches(object):
    """Restatement of MinibatchData (data.py:63-156).

    Holds a row-permuted copy of ``array``; iterating yields VIEWS of
    floor(N / minibatch) consecutive row blocks (tail rows are skipped);
    ``shuffle(index)`` gathers rows cumulatively; ``array`` un-shuffles.
    """

    def __init__(self, array, minibatch):
        if len(array) < minibatch:
            raise ValueError('Minibatch size should be smaller than the total '
                             'size. Given {} < {}'.format(len(array),
                                                          minibatch))
        self._a = array
        self.minibatch = minibatch
        self.order = np.arange(len(array))

    @property
    def dtype(self):
        return self._a.dtype

    @property
    def shape(self):
        return self._a.shape

    @property
    def n_loop(self):
        return int(len(self._a) / self.minibatch)

    def shuffle(self, index):
        self._a = self._a[index]
        self.order = self.order[index]

    @property
    def array(self):
        return self._a[self.order.argsort()]

    def __iter__(self):
        for r in range(self.n_loop):
            yield self._a[r * self.minibatch:(r + 1) * self.minibatch]


class Nones(object):
    """data.py:316-330 : an endless stream of None."""

    def shuffle(self, index):
        pass

    def __iter__(self):
        while True:
            yield None
