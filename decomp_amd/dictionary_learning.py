"""Online dictionary learning with lasso regularisation -- drop-in for
``decomp.dictionary_learning`` on MI355X.

Same entry point, argument meaning, return convention and error behaviour as the
reference's decomp/dictionary_learning.py:12-168 (``method='block_cd'``, Mairal et al.).
The epoch / minibatch loop and the seeded shuffle stay in Python (host logic, as in the
reference: the permutation comes from ``np.random.RandomState(random_seed)`` so the
minibatch composition is identical); each minibatch step -- LASSO inner solve, the
A / B statistics, the sequential atom sweep and max|D - D_new| -- is one call into
libdecomp_hip.so (``dcp_dict_step_*``, decomp_amd/csrc/dict_impl.hpp).

In-core data (y, x on the device; the BASELINE configs) take ``solve_cd_indexed``: y is never
permuted or copied -- the reference's cumulative shuffle (utils/data.py:152-156: a full gather of
y and x per epoch) becomes an index, each step gathers only ITS minibatch rows into a reused
staging block (the rows of step s+1 beside step s's atom sweep, where the chip is nearly
idle: ``dcp_dict_prefetch_rows_bytes``), the stop test
``max|D - D_new| < tol`` (dictionary_learning.py:161-162) is read one step late so the host never
idles the GPU, and a speculative step after a passed test is discarded.  Same minibatch
composition, same arithmetic per step, same return values.

``mask`` selects the masked variant (solve_cd_mask, dictionary_learning.py:171-231): same
structure with a per-channel [K, F, K] Gram statistic (``dcp_dict_mask_step_*``; a parity
path, single GPU, memory K*F*K elements as in the reference).
"""
import ctypes
import os

import numpy as np

from . import _arrays, _hip, lasso
from ._arrays import get_array_module
from .utils import assertion
from .utils.data import MinibatchData, AsyncMinibatchData, NoneIterator

_JITTER = 1.0e-15


def solve(y, D, alpha, x=None, tol=1.0e-3,
          minibatch=None, maxiter=1000, method='block_cd',
          lasso_method='cd', lasso_iter=10, lasso_tol=1.0e-5,
          mask=None, random_seed=None):
    """
    Learn a dictionary with lasso regularisation,
        argmin_{x, D} |y - xD|^2 + alpha |x|   s.t. |D_j|^2 <= 1,
    y: [n_samples, n_channels], x: [n_samples, n_features], D: [n_features, n_channels];
    float or complex, NumPy (results returned as NumPy) or torch CUDA tensors.
    ``minibatch`` is required.  Returns (it, D, x) as the reference does.
    """
    import torch
    kind = get_array_module(D)
    x_given = x
    if x is None:    # QUIRK: ones in D's dtype (dictionary_learning.py:58-59), not zeros
        x = lasso._ZerosLike((y.shape[0], D.shape[0]), _arrays.np_dtype(D))

    assertion.assert_dtypes(y=y, D=D, x=x)                               # :65-69
    assertion.assert_dtypes(mask=mask, dtypes='f')
    assertion.assert_shapes('x', x, 'D', D, axes=1)
    assertion.assert_shapes('y', y, 'D', D, axes=[-1])
    assertion.assert_shapes('y', y, 'mask', mask)

    if minibatch is None:                                                # :72-74
        raise NotImplementedError('Only online methods are implemented. '
                                  'minibatch is required.')
    if kind == 'numpy':
        get_array_module(y, D, x_given, mask)                            # :78
    if method != 'block_cd':                                             # :109-111
        raise NotImplementedError('Method %s is not yet implemented' % method)
    lasso._dict_method_code(lasso_method)      # NotImplementedError for unknown solvers
    assert _arrays.np_dtype(D).kind != 'c' or not lasso_method.endswith('_pos')   # lasso.py:92

    Dd = _arrays.to_device(D, copy=True)
    dev = Dd.device.index
    rdt = torch.float32 if Dd.dtype in (torch.float32, torch.complex64) else torch.float64

    def dataset(a, needs_update, real=False):
        """dictionary_learning.py:87-97: with a device D, NumPy arrays stay in pinned host
        memory and are streamed (AsyncMinibatchData); everything else is in-core."""
        if kind == 'torch' and not _arrays.is_torch(a):
            if real:
                a = np.ascontiguousarray(a, dtype=_arrays._torch_to_np_dtype(rdt))
            return AsyncMinibatchData(a, minibatch, needs_update=needs_update, device=dev)
        t = _arrays.to_device(a, dev, copy=needs_update)
        if real and t.dtype != rdt:
            t = t.to(rdt)
        return MinibatchData(t.contiguous(), minibatch)

    streamed = kind == 'torch' and (not _arrays.is_torch(y) or
                                    (x_given is not None and not _arrays.is_torch(x_given)))
    if mask is None and not streamed:
        yd = _arrays.to_device(y, dev)
        if x_given is None:
            xd = torch.ones(x.shape, dtype=Dd.dtype, device=Dd.device)
        else:
            xd = _arrays.to_device(x_given, dev, copy=True)
        rng = np.random.RandomState(random_seed)                         # :85
        it, Dout, xout = solve_cd_indexed(yd, Dd, alpha, xd, tol, minibatch, maxiter,
                                          lasso_method, lasso_iter, lasso_tol, rng)
        return it, _arrays.to_caller(Dout, kind), _arrays.to_caller(xout, kind)

    ybat = dataset(y, False)                                             # :79-80
    if x_given is None:
        xbat = MinibatchData(torch.ones(x.shape, dtype=Dd.dtype, device=Dd.device), minibatch)
    else:
        xbat = dataset(x_given, True)
    rng = np.random.RandomState(random_seed)                             # :85
    if mask is None:
        it, Dout, xout = solve_cd(ybat, Dd, alpha, xbat, tol, minibatch, maxiter,
                                  lasso_method, lasso_iter, lasso_tol, rng, kind)
    else:
        mbat = dataset(mask, False, real=True)
        it, Dout, xout = solve_cd_mask(ybat, Dd, alpha, xbat, tol, minibatch, maxiter,
                                       lasso_method, lasso_iter, lasso_tol, rng, kind, mbat)
    return it, _arrays.to_caller(Dout, kind), _arrays.to_caller(xout, kind)


def solve_cd_indexed(y, D, alpha, x, tol, minibatch, maxiter,
                     lasso_method, lasso_iter, lasso_tol, rng):
    """dictionary_learning.py:114-168 for in-core data: ``y`` [N, F] and ``x`` [N, K] are device
    arrays in their ORIGINAL row order (x is updated in place, y is only read), ``D`` a device
    array (normalised in place).  Returns (it, D, x)."""
    import torch
    import time as _time
    t_fn0 = _time.perf_counter()
    N = y.shape[0]
    if N < minibatch:                                                    # utils/data.py:79-82
        raise ValueError('Minibatch size should be smaller than the total '
                         'size. Given {} < {}'.format(N, minibatch))
    K, F = D.shape
    sfx = _arrays.suffix(D)
    lib, h = _arrays.lib_handle(D)
    step = getattr(lib, 'dcp_dict_step_async_' + sfx)
    code = lasso._dict_method_code(lasso_method)
    pcd_table = lasso._dict_pcd_table(lasso_method, K, lasso_iter, D)    # noqa: F841 (kept alive)
    dev = D.device
    rdt = torch.float32 if D.dtype in (torch.float32, torch.complex64) else torch.float64
    A = torch.zeros((K, K), dtype=D.dtype, device=dev)                   # :122-123
    B = torch.zeros((K, F), dtype=D.dtype, device=dev)
    _arrays.l2_normalize_(D, strict=True)                                # :126
    D_new = torch.empty_like(D)
    n_loop = int(N / minibatch)                                          # utils/data.py:107-108
    y_stage = [torch.empty((minibatch, F), dtype=y.dtype, device=dev) for _ in range(2)]
    x_stage = torch.empty((minibatch, K), dtype=x.dtype, device=dev)
    md_host = _pinned(torch, 'md', (2,), rdt)
    md_np = md_host.numpy()                  # the same two words, read without a torch dispatch

    def wait_maxdiff(slot):
        # max|D - D_new| >= 0 (or NaN): the slot holds the sentinel -1 from just before its step was enqueued
        # until the step's last kernel stores the value.  Polling instead of an event: the event's barrier packet
        # (system-scope release) idles the GPU for ~6 us behind every step.
        spins = 0
        while md_np[slot] == -1.0:
            spins += 1
            if (spins & 63) == 0:
                _time.sleep(0)               # hand the GIL over: the helper thread draws the next epoch's order
            if spins > 2000000:              # ~0.3 s: fall back to a blocking wait
                main.synchronize()
                break
        return float(md_np[slot])

    # two preallocated pinned blocks for the epochs' row orders, filled by a plain single-threaded copy
    # (Tensor.pin_memory() per epoch would run torch's parallel CPU copy: on a many-core host its worker
    # threads spin after every call and starve the launching thread of its CPU share -- measured: 50-90 ms
    # stalls of single launches under the container's CPU quota)
    host_blocks = [[_pinned(torch, 'idx%d' % b, (n_loop * minibatch,), torch.int64), None] for b in range(2)]
    main = torch.cuda.current_stream(dev)
    side = _side_stream(torch, dev)
    side.wait_stream(main)
    row_bytes_y = F * y.element_size()
    row_bytes_x = K * x.element_size()
    from .utils.data import _move_rows

    def next_order(state):
        """The next epoch's row order (host only): position p of epoch e holds original row order_e[p],
        order_e = order_{e-1}[index] with the index shuffled in place once per epoch (utils/data.py:152-156,
        dictionary_learning.py:131-133), written into that epoch's pinned block."""
        index, order, block = state
        rng.shuffle(index)
        order = order[index]
        block.numpy()[...] = order[:n_loop * minibatch]     # plain single-threaded copy (see host_blocks)
        return order

    def steps():
        """(epoch number, device index of the minibatch's ORIGINAL rows, upload event), in the reference's
        order.  The permutation of epoch e + 1 is drawn by a helper thread while epoch e computes (the main
        thread spends its time inside the library with the GIL released): ~1.5 ms of host work per epoch at
        65536 rows that would otherwise sit between two steps with the GPU idle.  The RNG is touched by one
        thread at a time and in the reference's order."""
        index = np.arange(N)                                             # :120
        order = np.arange(N)
        pool = _helper_pool() if maxiter > 2 else None
        try:
            fut = None
            for it in range(1, maxiter):                                     # :130
                hb = host_blocks[it & 1]
                if fut is None:
                    if hb[1] is not None:
                        hb[1].synchronize()
                    order = next_order((index, order, hb[0]))
                else:
                    order = fut.result()
                with torch.cuda.stream(side):
                    idx_dev = hb[0].to(dev, non_blocking=True)
                    up = torch.cuda.Event()
                    up.record(side)
                hb[1] = up
                # allocated in the side stream's pool, read by the compute stream's x gather / scatter too: the
                # allocator must not hand the block to the next epoch's upload while those are still queued
                idx_dev.record_stream(main)
                fut = None
                if pool is not None and it + 1 < maxiter:
                    nb = host_blocks[(it + 1) & 1]
                    if nb[1] is not None:
                        nb[1].synchronize()          # the upload that last read that block (the epoch before)
                    fut = pool.submit(next_order, (index, order, nb[0]))
                for m in range(n_loop):
                    yield it, idx_dev[m * minibatch:(m + 1) * minibatch], up, hb[0]
        finally:
            if fut is not None:
                fut.result()         # the helper must not outlive the call (it writes a pinned block of it)

    def fetch_y(idx, buf, stream):
        with torch.cuda.stream(stream):
            _move_rows('dcp_gather_rows_bytes', y, idx, minibatch, row_bytes_y, y_stage[buf])

    trace = os.environ.get('DCP_DL_TRACE', '0') == '1'      # analysis knob: host seconds per phase
    tacc = {'next_epoch_or_step': 0.0, 'gathers': 0.0, 'step_call': 0.0, 'stop_test_wait': 0.0, 'scatter': 0.0}
    lasso_it = ctypes.c_int(0)
    prefetch = os.environ.get('DCP_DL_PREFETCH', '1') != '0'     # analysis knob: gather on the compute stream
    t_loop0 = _time.perf_counter()
    gen = steps()
    cur = next(gen, None)
    t_first = _time.perf_counter()
    count = 0
    pending = None          # (slot, it, D_new) of the step before
    ready = False           # y_stage[count & 1] already holds the current step's rows (prefetched)
    try:
        while cur is not None:
            it, idx, uploaded = cur[0], cur[1], cur[2]
            buf = count & 1
            main.wait_event(uploaded)
            if not ready:
                fetch_y(idx, buf, main)
            t0 = _time.perf_counter()
            nxt = next(gen, None)
            t1 = _time.perf_counter()
            tacc['next_epoch_or_step'] += t1 - t0
            ready = False
            if nxt is not None and prefetch:
                # rows of the NEXT step: registered with the library, which runs the gather beside this step's
                # atom sweep (after its statistics product, the last reader of y_stage) and joins it before the
                # step's last kernel -- the chip and its HBM are nearly idle there
                main.wait_event(nxt[2])             # that minibatch's index upload
                lib, h = _arrays.lib_handle(D)
                _hip.check(h, lib.dcp_dict_prefetch_rows_bytes(h, _arrays.ptr(y), _arrays.ptr(nxt[1]), minibatch,
                                                               row_bytes_y, _arrays.ptr(y_stage[buf ^ 1])),
                           'dcp_dict_prefetch_rows_bytes')
                ready = True
            _move_rows('dcp_gather_rows_bytes', x, idx, minibatch, row_bytes_x, x_stage)
            t2 = _time.perf_counter()
            tacc['gathers'] += t2 - t1
            theta_plus1 = count * minibatch + 1.0                        # :143-144
            beta = (theta_plus1 - minibatch) / theta_plus1
            lib, h = _arrays.lib_handle(D)
            slot = count & 1
            # max|D - D_new| is written by the step's last kernel straight into pinned host memory (device-mapped: the
            # same pointer serves the GPU), no copy kernel behind the step
            md_np[slot] = -1.0
            rc = step(h, _arrays.ptr(y_stage[buf]), _arrays.ptr(x_stage), _arrays.ptr(D), _arrays.ptr(D_new),
                      _arrays.ptr(A), _arrays.ptr(B), minibatch, F, K, float(beta), float(alpha), code,
                      int(lasso_iter), float(lasso_tol), _arrays.ptr(md_host[slot:slot + 1]),
                      ctypes.byref(lasso_it))
            _hip.check(h, rc, 'dcp_dict_step_async_' + sfx)
            t3 = _time.perf_counter()
            tacc['step_call'] += t3 - t2
            # stop test of the PREVIOUS step (:161-162), now that this one is enqueued
            if pending is not None:
                pslot, pit, pD = pending
                if wait_maxdiff(pslot) < tol:
                    # this step ran speculatively on the converged dictionary: its codes are not
                    # scattered, its A / B / D_new are dropped
                    return pit, pD, x
            t4 = _time.perf_counter()
            tacc['stop_test_wait'] += t4 - t3
            _move_rows('dcp_scatter_rows_bytes', x_stage, idx, minibatch, row_bytes_x, x)
            tacc['scatter'] += _time.perf_counter() - t4
            pending = (slot, it, D_new)
            D, D_new = D_new, D
            count += 1
            cur = nxt
    except KeyboardInterrupt:                                            # :166-167
        torch.cuda.synchronize(dev)
        return (pending[1] if pending else 1), D, x
    finally:
        if trace:
            print('setup ms %.3f first permutation + upload ms %.3f loop total ms %.3f' %
                  (1e3 * (t_loop0 - t_fn0), 1e3 * (t_first - t_loop0), 1e3 * (_time.perf_counter() - t_loop0)), flush=True)
            print('solve_cd_indexed host ms per step:', {k: round(1e3 * v / max(count, 1), 4) for k, v in tacc.items()},
                  'steps', count, flush=True)
        side.synchronize()      # the index uploads: nothing of this call is left on the side stream
        # a registration the step never consumed (an error between the two calls) must not outlive the staging blocks
        lib.dcp_dict_prefetch_rows_bytes(_hip.handle(dev.index), None, None, 0, 0, None)
    if pending is not None:
        pslot, pit, pD = pending
        if wait_maxdiff(pslot) < tol:
            return pit, pD, x
    return maxiter, D, x


_SIDE_STREAMS = {}
_PINNED = {}
_POOL = []


def _pinned(torch, tag, shape, dtype):
    """Pinned host blocks are kept between calls (hipHostMalloc costs a good fraction of a millisecond, a whole
    minibatch step's worth per solve): one per (tag, shape, dtype); a call is not re-entrant per process anyway
    (one library handle per device)."""
    key = (tag, tuple(shape), dtype)
    t = _PINNED.get(key)
    if t is None:
        if len(_PINNED) > 16:
            _PINNED.clear()
        t = torch.zeros(shape, dtype=dtype, pin_memory=True)
        _PINNED[key] = t
    return t


def _helper_pool():
    if not _POOL:
        import concurrent.futures
        _POOL.append(concurrent.futures.ThreadPoolExecutor(max_workers=1))
    return _POOL[0]


def _side_stream(torch, dev):
    st = _SIDE_STREAMS.get(dev.index)
    if st is None:
        st = torch.cuda.Stream(device=dev)
        _SIDE_STREAMS[dev.index] = st
    return st


def solve_cd(y, D, alpha, x, tol, minibatch, maxiter,
             lasso_method, lasso_iter, lasso_tol, rng, xp=None):
    """dictionary_learning.py:114-168 with device arrays: ``y`` / ``x`` are
    decomp_amd.utils.data.MinibatchData, ``D`` a device array (normalised in place)."""
    import torch
    K, F = D.shape
    sfx = _arrays.suffix(D)
    lib, h = _arrays.lib_handle(D)
    step = getattr(lib, 'dcp_dict_step_' + sfx)
    code = lasso._dict_method_code(lasso_method)
    pcd_table = lasso._dict_pcd_table(lasso_method, K, lasso_iter, D)    # noqa: F841 (kept alive)
    index = np.arange(y.size)                                            # :120
    A = torch.zeros((K, K), dtype=D.dtype, device=D.device)              # :122-123
    B = torch.zeros((K, F), dtype=D.dtype, device=D.device)
    _arrays.l2_normalize_(D, strict=True)                                # :126
    D_new = torch.empty_like(D)
    maxdiff = ctypes.c_double(0.0)
    lasso_it = ctypes.c_int(0)
    count = 0
    for it in range(1, maxiter):                                         # :130
        rng.shuffle(index)
        y.shuffle(index)
        x.shuffle(index)
        try:
            for y_mb, x_mb in zip(y, x):
                theta_plus1 = count * minibatch + 1.0                    # :143-144
                beta = (theta_plus1 - minibatch) / theta_plus1
                lib, h = _arrays.lib_handle(D)
                rc = step(h, _arrays.ptr(y_mb), _arrays.ptr(x_mb), _arrays.ptr(D),
                          _arrays.ptr(D_new), _arrays.ptr(A), _arrays.ptr(B),
                          y_mb.shape[0], F, K, float(beta), float(alpha), code,
                          int(lasso_iter), float(lasso_tol), ctypes.byref(maxdiff),
                          ctypes.byref(lasso_it))
                _hip.check(h, rc, 'dcp_dict_step_' + sfx)
                if maxdiff.value < tol:                                  # :161-162
                    return it, D_new, x.array
                D, D_new = D_new, D
                count += 1
        except KeyboardInterrupt:                                        # :166-167
            return it, D, x.array
    return maxiter, D, x.array


def solve_cd_mask(y, D, alpha, x, tol, minibatch, maxiter,
                  lasso_method, lasso_iter, lasso_tol, rng, xp, mask):
    """dictionary_learning.py:171-231 with device arrays (mask: MinibatchData)."""
    import torch
    K, F = D.shape
    sfx = _arrays.suffix(D)
    lib, h = _arrays.lib_handle(D)
    step = getattr(lib, 'dcp_dict_mask_step_' + sfx)
    code = lasso._dict_method_code(lasso_method)
    pcd_table = lasso._dict_pcd_table(lasso_method, K, lasso_iter, D)    # noqa: F841 (kept alive)
    index = np.arange(y.size)
    A = torch.zeros((K, F, K), dtype=D.dtype, device=D.device)           # :179
    B = torch.zeros((K, F), dtype=D.dtype, device=D.device)
    _arrays.l2_normalize_(D, strict=True)                                # :183
    D_new = torch.empty_like(D)
    maxdiff = ctypes.c_double(0.0)
    lasso_it = ctypes.c_int(0)
    count = 0
    for it in range(1, maxiter):
        rng.shuffle(index)
        y.shuffle(index)
        x.shuffle(index)
        mask.shuffle(index)
        try:
            for y_mb, x_mb, m_mb in zip(y, x, mask):
                theta_plus1 = count * minibatch + 1.0                    # :201-202
                beta = (theta_plus1 - minibatch) / theta_plus1
                lib, h = _arrays.lib_handle(D)
                rc = step(h, _arrays.ptr(y_mb), _arrays.ptr(m_mb), _arrays.ptr(x_mb), _arrays.ptr(D),
                          _arrays.ptr(D_new), _arrays.ptr(A), _arrays.ptr(B),
                          y_mb.shape[0], F, K, float(beta), float(alpha), code,
                          int(lasso_iter), float(lasso_tol), ctypes.byref(maxdiff),
                          ctypes.byref(lasso_it))
                _hip.check(h, rc, 'dcp_dict_mask_step_' + sfx)
                if maxdiff.value < tol:                                  # :225-226
                    return it, D_new, x.array
                D, D_new = D_new, D
                count += 1
        except KeyboardInterrupt:
            return it, D, x.array
    return maxiter, D, x.array
