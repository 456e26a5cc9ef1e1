"""Row-minibatch containers on the GPU (reference: decomp/utils/data.py:63-156,316-330).

``MinibatchData`` keeps a row-permuted copy of an array in device memory; iterating
yields VIEWS of floor(N / minibatch) consecutive row blocks (tail rows are skipped that
epoch), ``shuffle(index)`` gathers rows cumulatively (HIP gather kernel), ``array``
restores the original order.

``AsyncMinibatchData`` (reference: data.py:159-313) is the out-of-core variant: the array
stays in PINNED host memory and minibatches are staged through ``n_parallel`` device
buffers, each with its own HIP stream, so that the H2D copy of round r + n_parallel - 1
and the D2H write-back of round r - 1 overlap the kernels of round r.  288 GB of HBM holds
every BASELINE config in-core; this container is for data sets beyond that, and for the
reference's calling convention (host ``y`` / ``x`` with a device ``D``).
"""
import numpy as np

from .. import _arrays, _hip


def _gather_rows(t, index_dev):
    """t[index] for a [N, ...] device array (rows gathered by the HIP kernel)."""
    import torch
    t2 = t.reshape(t.shape[0], -1)
    out = torch.empty_like(t2)
    lib, h = _arrays.lib_handle(t2)
    _hip.check(h, lib.dcp_gather_rows_bytes(h, _arrays.ptr(t2), _arrays.ptr(index_dev), t2.shape[0],
                                            t2.shape[1] * t2.element_size(), _arrays.ptr(out)),
               'dcp_gather_rows_bytes')
    return out.reshape(t.shape)


class MinibatchData(object):
    def __init__(self, array, minibatch, shuffle_index=None):
        """array: device array, first axis = samples; minibatch: rows per block;
        shuffle_index: optional initial row permutation (data.py:128-145)."""
        self.minibatch = minibatch
        self._array = array
        self.size = array.shape[0]
        if self.size < self.minibatch:                        # data.py:79-82
            raise ValueError('Minibatch size should be smaller than the total '
                             'size. Given {} < {}'.format(self.size, self.minibatch))
        self.restore_index = np.arange(self.size)
        if shuffle_index is not None:
            self.shuffle(np.asarray(shuffle_index))

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


def _move_rows(fn_name, src, index_dev, rows, row_bytes, dst):
    """dcp_gather_rows_bytes / dcp_scatter_rows_bytes on torch's current stream."""
    lib, h = _arrays.lib_handle(index_dev)
    _hip.check(h, getattr(lib, fn_name)(h, _arrays.ptr(src), _arrays.ptr(index_dev), rows, row_bytes,
                                        _arrays.ptr(dst)), fn_name)


_COPY_STREAMS = {}


def _shared_copy_stream(torch, device):
    """ONE copy stream per device for every container.  HIP multiplexes streams onto a
    handful of hardware queues; a stream per staging buffer (as the reference creates,
    data.py:163-164) ends up sharing a queue with the compute stream, which serialises the
    copies with the kernels (seen in the rocprofv3 trace).  The link is the shared resource
    anyway, so one queue for all copies loses nothing."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    st = _COPY_STREAMS.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _COPY_STREAMS[key] = st
    return st


class _Slot(object):
    """One device staging buffer (data.py:159-183) and the stream its copies run on."""
    def __init__(self, torch, shape, dtype, device, use_stream):
        self.torch = torch
        self.buf = torch.empty(shape, dtype=dtype, device=device)
        self.stream = _shared_copy_stream(torch, device) if use_stream else None
        self.ready = None        # event: this buffer's last copy has completed

    def copy_stream(self):
        return self.stream if self.stream is not None else self.torch.cuda.current_stream(self.buf.device)

    def after_compute(self):
        """Order this slot's next copy after everything enqueued so far on the compute
        (current) stream -- the consumer's kernels on this buffer."""
        if self.stream is not None:
            self.stream.wait_stream(self.torch.cuda.current_stream(self.buf.device))

    def copied(self):
        """Mark the end of this buffer's copy on the copy stream."""
        if self.stream is not None:
            self.ready = self.torch.cuda.Event()
            self.ready.record(self.stream)

    def before_compute(self):
        """The compute stream waits (on the device, not the host) for THIS buffer's copy only,
        not for the prefetches queued behind it on the shared copy stream."""
        if self.stream is not None and self.ready is not None:
            self.torch.cuda.current_stream(self.buf.device).wait_event(self.ready)

    def synchronize(self):
        self.copy_stream().synchronize()


class AsyncMinibatchData(object):
    """Host-resident data streamed through the GPU minibatch by minibatch
    (reference: decomp/utils/data.py:214-313; same constructor arguments).

    Iterating yields DEVICE arrays [minibatch, ...] (torch tensors, valid until the next
    ``next()``); with ``needs_update`` the consumer's in-place changes are written back to
    the host array.  MI355X-first differences from the reference:
      * the host array is never permuted: it stays in the original row order in pinned
        (device-mapped) memory, ``shuffle`` only composes an index permutation, and each
        minibatch is GATHERED by a kernel straight out of host memory over PCIe
        (``dcp_gather_rows_bytes``; updated rows are scattered back the same way).  The
        reference permutes the whole host array on every shuffle (2 s per epoch at the
        BASELINE configs[2] size);
      * copies are ordered against the compute stream with device-side waits, the host
        never blocks inside an epoch;
      * the round that was yielded last is written back by ``flush()`` (called by
        ``array``, ``shuffle`` and ``__iter__``); in the reference a consumer that stops
        early -- ``zip(y, x, mask)`` stops at y -- loses x's last minibatch.
    """
    def __init__(self, array, minibatch, n_parallel=3, shuffle_index=None,
                 needs_update=True, use_stream=True, device=None):
        import torch
        from . import assertion
        self.torch = torch
        assertion.assert_shapes('array', array, 'shuffle_index', shuffle_index, axes=[0])
        array = np.asarray(array)
        self.minibatch = minibatch
        self.size = array.shape[0]
        if self.size < self.minibatch:                        # data.py:79-82
            raise ValueError('Minibatch size should be smaller than the total '
                             'size. Given {} < {}'.format(self.size, self.minibatch))
        if not torch.cuda.is_available():
            raise _hip.HipLibraryError('AsyncMinibatchData needs a GPU: decomp_amd computes '
                                       'only on the device.')
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        # pinned = page-locked AND mapped into the device's address space (hipHostMalloc)
        self._host = torch.empty(array.shape, dtype=torch.from_numpy(array[:0]).dtype,
                                 pin_memory=True)
        self._host.numpy()[...] = array                       # original row order, for good
        self._row_bytes = int(np.prod(array.shape[1:], dtype=np.int64)) * array.dtype.itemsize
        self.restore_index = np.arange(self.size)             # position -> original row
        if shuffle_index is not None:                         # data.py:245-250
            self.restore_index = self.restore_index[np.asarray(shuffle_index)]
        self._order_dev = None
        self.needs_update = needs_update
        self.n_parallel = int(min(n_parallel, self.n_loop))
        self._slots = [_Slot(torch, (minibatch,) + tuple(array.shape[1:]), self._host.dtype,
                             self.device, use_stream) for _ in range(self.n_parallel)]
        self.round = 0
        self._pending = -1          # round handed out and not yet written back

    # ---- the MinibatchBase surface (data.py:63-121) ----
    @property
    def shape(self):
        return tuple(self._host.shape)

    @property
    def dtype(self):
        return self._host.numpy().dtype

    @property
    def n_loop(self):
        return int(self.size / self.minibatch)

    def __len__(self):
        return self.size

    def _order(self, round_):
        """Device int64 indices (original rows) of the minibatch of this round."""
        if self._order_dev is None:
            self._order_dev = self.torch.from_numpy(
                np.ascontiguousarray(self.restore_index, dtype=np.int64)).to(self.device)
        return self._order_dev[round_ * self.minibatch:(round_ + 1) * self.minibatch]

    def _slot(self, round_):
        return self._slots[round_ % self.n_parallel]

    def _send(self, round_):
        slot = self._slot(round_)
        idx = self._order(round_)
        with self.torch.cuda.stream(slot.copy_stream()):
            _move_rows('dcp_gather_rows_bytes', self._host, idx, self.minibatch, self._row_bytes,
                       slot.buf)
        slot.copied()

    def _fetch(self, round_):
        slot = self._slot(round_)
        idx = self._order(round_)
        with self.torch.cuda.stream(slot.copy_stream()):
            _move_rows('dcp_scatter_rows_bytes', slot.buf, idx, self.minibatch, self._row_bytes,
                       self._host)

    def flush(self):
        """Write back the round that was handed out last and wait for every copy."""
        if self._pending >= 0:
            self._slot(self._pending).after_compute()
            if self.needs_update:
                self._fetch(self._pending)
            self._pending = -1
        for s in self._slots:
            s.synchronize()

    @property
    def array(self):
        """NumPy array in the ORIGINAL row order (data.py:267-270): the host array itself."""
        self.flush()
        return self._host.numpy()

    @property
    def _array(self):
        """The rows in the CURRENT (shuffled) order, as the reference's attribute of this name."""
        self.flush()
        return self._host.numpy()[self.restore_index]

    def shuffle(self, shuffle_index):
        """Cumulative row permutation (data.py:272-282): composes indices, moves no data."""
        from . import assertion
        if _arrays.is_torch(shuffle_index):
            shuffle_index = shuffle_index.cpu().numpy()
        shuffle_index = np.asarray(shuffle_index)
        assertion.assert_shapes('array', self._host, 'shuffle_index', shuffle_index, axes=[0])
        self.flush()
        self.restore_index = self.restore_index[shuffle_index]
        self._order_dev = None

    def __iter__(self):
        self.flush()
        self.round = 0
        for r in range(self.n_parallel):                      # data.py:284-290
            self._slot(r).after_compute()
            self._send(r)
        return self

    def __next__(self):
        return self.next()

    def next(self):
        if self.round > 0:                                    # data.py:292-299
            prev = self.round - 1
            self._slot(prev).after_compute()
            if self.needs_update:
                self._fetch(prev)
            self._pending = -1
            nxt = prev + self.n_parallel
            if nxt < self.n_loop:
                self._send(nxt)
        if self.round + 1 > self.n_loop:                      # data.py:301-305
            for s in self._slots:
                s.synchronize()
            raise StopIteration()
        slot = self._slot(self.round)
        slot.before_compute()
        self._pending = self.round
        self.round += 1
        return slot.buf


class NoneIterator(object):
    """An endless stream of None (data.py:316-330)."""
    def __iter__(self):
        while True:
            yield None

    def shuffle(self, index):
        pass
