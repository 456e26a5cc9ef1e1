"""Row-minibatch containers on the GPU (reference: decomp/utils/data.py:63-156,316-330).

``MinibatchData`` keeps a row-permuted copy of an array in device memory; iterating
yields VIEWS of floor(N / minibatch) consecutive row blocks (tail rows are skipped that
epoch), ``shuffle(index)`` gathers rows cumulatively (HIP gather kernel), ``array``
restores the original order.  The out-of-core ``AsyncMinibatchData`` of the reference is
not needed on a 288 GB device and is not provided.
"""
import numpy as np

from .. import _arrays, _hip


def _gather_rows(t, index_dev):
    """t[index] for a [N, ...] device array (rows gathered by the HIP kernel)."""
    import torch
    t2 = t.reshape(t.shape[0], -1)
    out = torch.empty_like(t2)
    lib, h = _arrays.lib_handle(t2)
    fn = getattr(lib, 'dcp_gather_rows_' + _arrays.suffix(t2))
    _hip.check(h, fn(h, _arrays.ptr(t2), _arrays.ptr(index_dev), t2.shape[0], t2.shape[1],
                     _arrays.ptr(out)), 'dcp_gather_rows')
    return out.reshape(t.shape)


class MinibatchData(object):
    def __init__(self, array, minibatch):
        """array: device array, first axis = samples; minibatch: rows per block."""
        self.minibatch = minibatch
        self._array = array
        self.size = array.shape[0]
        if self.size < self.minibatch:                        # data.py:79-82
            raise ValueError('Minibatch size should be smaller than the total '
                             'size. Given {} < {}'.format(self.size, self.minibatch))
        self.restore_index = np.arange(self.size)

    @property
    def shape(self):
        return tuple(self._array.shape)

    @property
    def dtype(self):
        return _arrays.np_dtype(self._array)

    @property
    def n_loop(self):
        return int(self.size / self.minibatch)

    @property
    def array(self):
        """The data in the ORIGINAL row order (data.py:147-150)."""
        return _gather_rows(self._array, self._index_dev(self.restore_index.argsort()))

    def _index_dev(self, index):
        import torch
        return torch.from_numpy(np.ascontiguousarray(index, dtype=np.int64)).to(self._array.device)

    def shuffle(self, shuffle_index):
        """Cumulative row permutation (data.py:152-156); shuffle_index: host int array."""
        if len(shuffle_index) != self.size:
            from .exceptions import ShapeMismatchError
            raise ShapeMismatchError('shuffle_index must have one entry per row')
        self._array = _gather_rows(self._array, self._index_dev(shuffle_index))
        self.restore_index = self.restore_index[shuffle_index]

    def __iter__(self):
        for r in range(self.n_loop):
            yield self._array[r * self.minibatch:(r + 1) * self.minibatch]


class NoneIterator(object):
    """An endless stream of None (data.py:316-330)."""
    def __iter__(self):
        while True:
            yield None

    def shuffle(self, index):
        pass
