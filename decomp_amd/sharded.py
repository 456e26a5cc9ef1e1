"""n_samples-sharded NMF multiplicative update: one process per GPU, rows of y / x / mask
partitioned across ranks, D replicated.

Per iteration each rank runs the x update and its share of the D-side sums on its own
rows (``dcp_nmf_mu_stats_*``), the [K, F+K] (or [K, 2F]) statistics are summed over
ranks with ONE all-reduce (RCCL over xGMI when the process group's backend is "nccl"),
and every rank applies the identical D update (``dcp_nmf_mu_update_*``).  Nothing else
is communicated; the stop test runs redundantly and identically on every rank.

The reference has no multi-device path; the single-process semantics reproduced here are
those of decomp/nmf_methods/batch_mu.py:8-26.

Round 4: under the "nccl" process-group backend the whole loop runs INSIDE the library
(``dcp_nmf_mu_sharded_*``): the handle owns an RCCL communicator (``attach_communicator``), the
all-reduce is enqueued on the solver's own stream between the two halves of the step and the host
only reads the lagged stop test -- no Python, no second stream, no event pair per step.  The Python
loop ``mu_loop`` stays as the executable statement of the host logic (it is what the gloo tests
drive, with an oracle-backed or HIP-backed step) and as the path for process groups RCCL cannot
serve (several ranks on one GPU).
"""
import ctypes
import os

from . import _arrays, _hip

_COMMS = {}     # device index -> (world, rank, id(group)) of the communicator its handle holds


def attach_communicator(device_tensor, group=None):
    """Give this rank's library handle an RCCL communicator spanning ``group`` (collective: every
    rank of the group must call it).  Rank 0 draws the unique id, torch.distributed carries its 128
    bytes to the others -- the only use of that channel -- and ``dcp_comm_init`` runs
    ncclCommInitRank on the handle's device.  Returns True when the in-library collective is usable
    on EVERY rank, False otherwise (gloo rehearsals with several ranks per GPU, librccl missing,
    ``DCP_SHARDED_LOOP=python``): the callers then keep the Python loop over torch.distributed."""
    import torch
    import torch.distributed as dist
    lib, h = _arrays.lib_handle(device_tensor)
    dev = device_tensor.device.index
    init = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if init else 1
    rank = dist.get_rank(group) if init else 0
    key = (world, rank, id(group))
    if _COMMS.get(dev) == key:
        return True
    if os.environ.get('DCP_SHARDED_LOOP', '') == 'python':
        return False
    if init and dist.get_backend(group) != 'nccl':
        # a process group RCCL cannot serve (gloo: several ranks may share one GPU).  The SAME in-library loops run
        # with the exchange handed to the library as a callback (dcp_comm_set_external): the statistics are staged
        # through pinned host memory and summed by torch.distributed on the CPU -- slow, for tests and rehearsals.
        if world == 1:
            return False
        return _attach_external(device_tensor, group, rank, world, key)
    ident = [None]
    if rank == 0:
        buf = ctypes.create_string_buffer(_hip.COMM_ID_BYTES)
        rc = lib.dcp_comm_unique_id(buf, _hip.COMM_ID_BYTES)
        ident = [buf.raw if rc == _hip.OK else b'']
    if world > 1:
        dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                   group=group)
    ok = 0
    if ident[0]:
        rc = lib.dcp_comm_init(h, ident[0], rank, world)
        ok = 1 if rc == _hip.OK else 0
    if world > 1:       # all ranks or none
        flag = torch.tensor([ok], dtype=torch.int32, device=device_tensor.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = int(flag.item())
    if not ok:
        lib.dcp_comm_destroy(h)
        _COMMS.pop(dev, None)
        return False
    _COMMS[dev] = key
    return True


_EXTERNAL = {}      # device index -> (callback object, staging tensors): kept alive while the handle uses them


def _attach_external(device_tensor, group, rank, world, key):
    import torch
    import torch.distributed as dist
    lib, h = _arrays.lib_handle(device_tensor)
    dev = device_tensor.device.index
    staging = {}

    def exchange(buf, count, dtype, stream, user):
        try:
            tdt = torch.float32 if dtype == 0 else torch.float64
            st = staging.get(dtype)
            if st is None or st.numel() < count:
                st = torch.empty((int(count),), dtype=tdt, pin_memory=True)
                staging[dtype] = st
            nbytes = int(count) * st.element_size()
            if lib.dcp_memcpy(h, ctypes.c_void_p(st.data_ptr()), ctypes.c_void_p(buf), nbytes) != _hip.OK:
                return 1
            torch.cuda.synchronize(dev)                        # everything up to the copy, on whatever stream
            dist.all_reduce(st[:int(count)], op=dist.ReduceOp.SUM, group=group)
            if lib.dcp_memcpy(h, ctypes.c_void_p(buf), ctypes.c_void_p(st.data_ptr()), nbytes) != _hip.OK:
                return 2
            return 0
        except Exception:       # never let an exception cross the C frame
            return 3
    cb = _hip.ALLREDUCE_FN(exchange)
    rc = lib.dcp_comm_set_external(h, ctypes.cast(cb, ctypes.c_void_p), None, rank, world)
    if rc != _hip.OK:
        return False
    _EXTERNAL[dev] = (cb, staging)
    _COMMS[dev] = key
    return True


def communicator_kind(device_tensor):
    """'rccl', 'external' (dcp_comm_set_external callback) or None for this device's handle."""
    dev = device_tensor.device.index
    if dev not in _COMMS:
        return None
    return 'external' if dev in _EXTERNAL else 'rccl'


def detach_communicator(device_tensor):
    lib, h = _arrays.lib_handle(device_tensor)
    _hip.check(h, lib.dcp_comm_destroy(h), 'dcp_comm_destroy')
    _COMMS.pop(device_tensor.device.index, None)
    _EXTERNAL.pop(device_tensor.device.index, None)


def comm_allreduce_(t):
    """In-place sum of a float / complex device tensor over the handle's communicator, on torch's
    current stream (``dcp_comm_allreduce_sum_*``)."""
    lib, h = _arrays.lib_handle(t)
    sfx = _arrays.suffix(t)
    n = t.numel() * (2 if sfx in ('c64', 'c128') else 1)
    fn = lib.dcp_comm_allreduce_sum_f32 if sfx in ('f32', 'c64') else lib.dcp_comm_allreduce_sum_f64
    _hip.check(h, fn(h, _arrays.ptr(t), n), 'dcp_comm_allreduce_sum')
    return t


def mu_solve_in_library(y, mask, x, D, lik, tol, maxiter):
    """``dcp_nmf_mu_sharded_*`` on this rank's rows (x and D updated in place).  Returns it."""
    lib, h = _arrays.lib_handle(D)
    sfx = _arrays.suffix(D)
    fn = getattr(lib, 'dcp_nmf_mu_sharded_' + sfx)
    it = ctypes.c_int(0)
    ctol = ctypes.c_float(tol) if sfx == 'f32' else ctypes.c_double(tol)
    _hip.check(h, fn(h, _arrays.ptr(y), _arrays.ptr(mask), _arrays.ptr(x), _arrays.ptr(D), y.shape[0],
                     y.shape[1], D.shape[0], lik, ctol, int(maxiter), ctypes.byref(it), None),
               'dcp_nmf_mu_sharded_' + sfx)
    return it.value


class HipStepBackend(object):
    """The two halves of one MU iteration on this rank's GPU, through the C ABI.

    x is double buffered: ``local_stats`` reads the current x and writes the next one, so
    the iteration that was enqueued speculatively (see ``mu_loop``) can be discarded with
    ``rollback``.  ``x`` is always the buffer holding the current iterate."""

    def __init__(self, y, mask, x, D, lik):
        import torch
        self.torch = torch
        self.y, self.mask, self.lik = y, mask, lik
        self.x = x
        self._x_other = torch.empty_like(x)
        self.N, self.F = y.shape
        self.K = D.shape[0]
        self.sfx = _arrays.suffix(D)
        lib = _hip.load()
        self.W = lib.dcp_nmf_mu_stats_width(self.F, self.K, lik, 0 if mask is None else 1)
        self.stats = torch.empty((self.K, self.W), dtype=D.dtype, device=D.device)
        self.maxdiff = torch.zeros((2,), dtype=D.dtype, device=D.device)
        self._host = torch.zeros((2,), dtype=D.dtype).pin_memory()
        self._events = [None, None]
        self._ym = self._bits = None
        if mask is not None:
            # loop-invariant mask work, once per run: y o mask and (float32, 0/1 mask) its row bits
            h = _arrays.lib_handle(D)[1]
            self._ym = torch.empty_like(y)
            words = lib.dcp_nmf_mask_bits_words(self.N, self.F)
            bits = torch.empty((words,), dtype=torch.int32, device=D.device) if self.sfx == 'f32' else None
            binary = ctypes.c_int(0)
            fn = getattr(lib, 'dcp_nmf_mask_prepare_' + self.sfx)
            _hip.check(h, fn(h, _arrays.ptr(y), _arrays.ptr(mask), self.N, self.F, _arrays.ptr(self._ym),
                             _arrays.ptr(bits), ctypes.byref(binary)), 'dcp_nmf_mask_prepare')
            self._bits = bits if binary.value else None

    def local_stats(self, D):
        lib, h = _arrays.lib_handle(D)
        if self.mask is not None:
            fn = getattr(lib, 'dcp_nmf_mu_stats_prepared_' + self.sfx)
            _hip.check(h, fn(h, _arrays.ptr(self._ym), _arrays.ptr(self.mask), _arrays.ptr(self._bits),
                             _arrays.ptr(self.x), _arrays.ptr(self._x_other), _arrays.ptr(D), self.N,
                             self.F, self.K, self.lik, _arrays.ptr(self.stats)), 'dcp_nmf_mu_stats_prepared')
        else:
            fn = getattr(lib, 'dcp_nmf_mu_stats_' + self.sfx)
            _hip.check(h, fn(h, _arrays.ptr(self.y), None, _arrays.ptr(self.x),
                             _arrays.ptr(self._x_other), _arrays.ptr(D), self.N, self.F, self.K,
                             self.lik, _arrays.ptr(self.stats)), 'dcp_nmf_mu_stats')
        self.x, self._x_other = self._x_other, self.x
        return self.stats

    def rollback(self):
        """Forget the last local_stats: x is again the iterate it started from."""
        self.x, self._x_other = self._x_other, self.x

    def update(self, stats, D, D_new, slot):
        """Enqueue the D update; its max|dD| lands asynchronously in host slot ``slot``."""
        lib, h = _arrays.lib_handle(D)
        fn = getattr(lib, 'dcp_nmf_mu_update_' + self.sfx)
        md = self.maxdiff[slot:slot + 1]            # zero on entry (ping-pong, see the C ABI)
        nxt = self.maxdiff[(slot ^ 1):(slot ^ 1) + 1]
        _hip.check(h, fn(h, _arrays.ptr(stats), _arrays.ptr(D), _arrays.ptr(D_new), self.F,
                         self.K, self.lik, 0 if self.mask is None else 1, _arrays.ptr(md),
                         _arrays.ptr(nxt)), 'dcp_nmf_mu_update')
        self._host[slot:slot + 1].copy_(md, non_blocking=True)
        ev = self.torch.cuda.Event()
        ev.record()
        self._events[slot] = ev

    def read_maxdiff(self, slot):
        self._events[slot].synchronize()
        return float(self._host[slot])


def mu_loop(backend, D, tol, maxiter, group=None, world_size=1, new_like=None):
    """batch_mu.py:16-26 with the statistics all-reduced over ``group``.

    backend: local_stats(D) -> stats array (advances backend.x), rollback(),
             update(stats, D, D_new, slot), read_maxdiff(slot) -> float
             (HipStepBackend; the CPU tests of this host logic inject an oracle-backed
             stand-in).
    Returns (it, D); backend.x is the matching x.  The stop test of iteration i
    (batch_mu.py:22) is read AFTER iteration i+1 has been enqueued, so neither the GPU
    nor the collective ever waits for the host; if iteration i did converge, the
    speculative iteration i+1 is rolled back.  Every rank sees the same all-reduced
    statistics, hence takes the same decision at the same iteration.
    """
    import torch.distributed as dist
    D_new = new_like(D)
    for it in range(1, maxiter):
        stats = backend.local_stats(D)
        if world_size > 1:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        backend.update(stats, D, D_new, it & 1)
        if it > 1 and backend.read_maxdiff((it - 1) & 1) < tol:
            backend.rollback()              # discard iteration `it`
            return it - 1, D                # D is D_new of iteration it-1
        D, D_new = D_new, D
    if maxiter > 1 and backend.read_maxdiff((maxiter - 1) & 1) < tol:
        return maxiter - 1, D
    return maxiter, D


def nmf_solve_sharded(y_local, D, x_local=None, tol=1.0e-3, maxiter=1000, likelihood='l2',
                      mask_local=None, group=None):
    """``decomp.nmf.solve(method='mu')`` for a row-sharded problem.

    Every rank passes its own rows (torch CUDA tensors) and the same D.  Returns
    (it, D, x_local); D and it are identical on all ranks.  torch.distributed must be
    initialised (backend "nccl" = RCCL on ROCm) unless the world size is 1.
    """
    import torch
    import torch.distributed as dist
    from . import nmf as _nmf
    from .utils import assertion
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    assertion.assert_dtypes(y=y_local, D=D, x=x_local, mask=mask_local, dtypes='f')
    assertion.assert_shapes('y', y_local, 'D', D, axes=[-1])
    assertion.assert_shapes('y', y_local, 'mask', mask_local)
    lik = _nmf._likelihood_code(likelihood)
    y = _arrays.to_device(y_local)
    m = _arrays.to_device(mask_local)
    Dd = _arrays.to_device(D, copy=True)
    if x_local is None:
        x = torch.ones((y.shape[0], Dd.shape[0]), dtype=Dd.dtype, device=Dd.device)
    else:
        x = _arrays.to_device(x_local, copy=True)
        assertion.assert_shapes('x', x, 'D', D, axes=1)
    assertion.assert_nonnegative(Dd)
    assertion.assert_nonnegative(x)
    if likelihood in ['kl']:
        assertion.assert_nonnegative(y)
    _arrays.l2_normalize_(Dd, strict=True)
    if world > 1 and attach_communicator(Dd, group):
        # the shipped multi-GPU path: the whole loop, collective included, behind the C ABI
        it = mu_solve_in_library(y, m, x, Dd, lik, tol, maxiter)
        return it, Dd, x
    backend = HipStepBackend(y, m, x, Dd, lik)
    it, Dout = mu_loop(backend, Dd, tol, maxiter, group=group, world_size=world,
                       new_like=torch.empty_like)
    return it, Dout, backend.x


# ======================================================================================
# Online dictionary learning, minibatch rows split over the ranks
# ======================================================================================
class HipDictBackend(object):
    """The two halves of one dictionary-learning minibatch step on this rank's GPU."""

    def __init__(self, D, lasso_method, lasso_iter, lasso_tol, alpha):
        import torch
        from . import lasso
        self.torch = torch
        self.sfx = _arrays.suffix(D)
        self.K, self.F = D.shape
        self.code = lasso._dict_method_code(lasso_method)
        self._pcd_table = lasso._dict_pcd_table(lasso_method, self.K, lasso_iter, D)
        self.lasso_iter, self.lasso_tol, self.alpha = int(lasso_iter), float(lasso_tol), float(alpha)
        self.stats = torch.empty((self.K, self.F + self.K), dtype=D.dtype, device=D.device)
        rdt = torch.float32 if D.dtype in (torch.float32, torch.complex64) else torch.float64
        self.md = torch.zeros((1,), dtype=rdt, device=D.device)
        self._host_md = None
        self.lasso_it = ctypes.c_int(0)

    def local_stats(self, y_rows, x_rows, D):
        """x_rows <- lasso(y_rows, D) in place; returns this rank's x^H [y | x]."""
        lib, h = _arrays.lib_handle(D)
        fn = getattr(lib, 'dcp_dict_stats_' + self.sfx)
        _hip.check(h, fn(h, _arrays.ptr(y_rows), _arrays.ptr(x_rows), _arrays.ptr(D), y_rows.shape[0],
                         self.F, self.K, self.alpha, self.code, self.lasso_iter, self.lasso_tol,
                         _arrays.ptr(self.stats), ctypes.byref(self.lasso_it)), 'dcp_dict_stats')
        return self.stats

    def update(self, stats, beta, A, B, D, D_new):
        """A, B accumulation + atom sweep + max|D - D_new| (returned as a float)."""
        lib, h = _arrays.lib_handle(D)
        fn = getattr(lib, 'dcp_dict_update_' + self.sfx)
        _hip.check(h, fn(h, _arrays.ptr(stats), float(beta), _arrays.ptr(A), _arrays.ptr(B), _arrays.ptr(D),
                         _arrays.ptr(D_new), self.F, self.K, _arrays.ptr(self.md)), 'dcp_dict_update')
        return float(self.md.item())

    def update_async(self, stats, beta, A, B, D, D_new):
        """Enqueue the A/B accumulation + atom sweep; max|D - D_new| stays on the device."""
        lib, h = _arrays.lib_handle(D)
        fn = getattr(lib, 'dcp_dict_update_' + self.sfx)
        _hip.check(h, fn(h, _arrays.ptr(stats), float(beta), _arrays.ptr(A), _arrays.ptr(B), _arrays.ptr(D),
                         _arrays.ptr(D_new), self.F, self.K, _arrays.ptr(self.md)), 'dcp_dict_update')

    def maxdiff_token(self, slot):
        """Start the copy of the device max|dD| into pinned host slot ``slot`` (0 / 1) and return a
        token for ``read_maxdiff`` -- read one step later, when the next step is already enqueued."""
        if self._host_md is None:
            self._host_md = self.torch.zeros((2,), dtype=self.md.dtype).pin_memory()
        self._host_md[slot:slot + 1].copy_(self.md, non_blocking=True)
        ev = self.torch.cuda.Event()
        ev.record()
        return (ev, slot)

    def read_maxdiff(self, token):
        ev, slot = token
        ev.synchronize()
        return float(self._host_md[slot])

    def gather(self, src, index_dev, n, out):
        """out[:n] = src[index[:n]] (rows; HIP row mover)."""
        if n:
            from .utils import data as _data
            _data._move_rows('dcp_gather_rows_bytes', src, index_dev, n, src.shape[1] * src.element_size(), out)

    def scatter(self, src, index_dev, n, out):
        """out[index[:n]] = src[:n]."""
        if n:
            from .utils import data as _data
            _data._move_rows('dcp_scatter_rows_bytes', src, index_dev, n, src.shape[1] * src.element_size(), out)


def dict_loop(backend, y_local, x_local, row0, n_total, D, tol, minibatch, maxiter, rng, new_like, zeros,
              index_to_device, group=None, world_size=1, allreduce=None):
    """dictionary_learning.py:114-168 with the SAMPLES sharded: rank r owns the original rows
    [row0, row0 + len(y_local)) of y and x for the whole run -- nothing but the [K, F+K] statistics
    ever crosses ranks, ONE all-reduce per minibatch step (north_star).

    Every rank draws the same permutation stream from the shared ``RandomState`` (the reference's
    cumulative shuffle, data.py:152-156), so minibatch m of an epoch is the same SET of original rows
    as in the single-process run; each rank processes the members of that set it owns (its share
    varies around minibatch / world_size), gathers them into a staging block, runs the LASSO and
    x^H [y | x] on it (``dcp_dict_stats_*``), scatters the new codes back into its own x, and the
    summed statistics drive the identical, redundant A/B update + atom sweep on every rank.
    ``allreduce``: in-place sum of the statistics over the ranks; None = torch.distributed.all_reduce on
    ``group`` (gloo rehearsals), ``comm_allreduce_`` = RCCL on the library handle's own stream.
    max|D - D_new| is read one step late (as the NMF loop does): the next minibatch is already
    enqueued when the host looks at it, and is discarded if the test had passed.

    Deviations from the single-process run, both at rounding / tolerance level: the statistics are
    summed in a different order; the LASSO's early exit (|dx| < lasso_tol on iterations 0, 10, ...,
    lasso.py:293) is taken per rank on its own rows instead of on the whole minibatch, so when it fires
    on one rank and not on another the codes differ by what the remaining iterations would have moved
    them, i.e. by O(lasso_tol) (tests/test_gpu_dictionary.py::test_sharded_dictionary_early_exit_per_rank).
    Returns (it, D, x_local)."""
    import torch.distributed as dist
    import numpy as np
    import torch
    K, F = D.shape
    A = zeros((K, K))
    B = zeros((K, F))
    D_new = new_like(D)
    n_local = y_local.shape[0]
    cap = min(n_local, minibatch)
    y_stage = torch.empty((max(cap, 1), F), dtype=y_local.dtype, device=y_local.device)
    x_stage = torch.empty((max(cap, 1), K), dtype=x_local.dtype, device=x_local.device)
    order = np.arange(n_total)                  # position -> original row, composed over epochs
    index = np.arange(n_total)
    n_loop = int(n_total / minibatch)
    if n_total < minibatch:
        raise ValueError('Minibatch size should be smaller than the total '
                         'size. Given {} < {}'.format(n_total, minibatch))
    count = 0
    pending = None                              # (max|dD| token, it, D_new) of the step before
    step = 0
    for it in range(1, maxiter):
        rng.shuffle(index)                      # dictionary_learning.py:131-133
        order = order[index]
        # this rank's members of every minibatch of the epoch: ONE index upload per epoch, sliced on the device
        mines = []
        for m in range(n_loop):
            rows = order[m * minibatch:(m + 1) * minibatch]
            mines.append(rows[(rows >= row0) & (rows < row0 + n_local)] - row0)
        offs = np.concatenate([[0], np.cumsum([len(v) for v in mines])]).astype(np.int64)
        idx_all = index_to_device(np.concatenate(mines)) if offs[-1] else None
        for m in range(n_loop):
            n = int(offs[m + 1] - offs[m])
            idx = idx_all[int(offs[m]):int(offs[m + 1])] if n else None
            backend.gather(y_local, idx, n, y_stage)
            backend.gather(x_local, idx, n, x_stage)
            if n:
                stats = backend.local_stats(y_stage[:n], x_stage[:n], D)
            else:                               # this rank owns no row of the minibatch
                stats = backend.stats.zero_()
            if world_size > 1:                                                # the ONLY collective
                if allreduce is not None:       # RCCL on the handle's own stream (dcp_comm_allreduce_sum_*)
                    allreduce(stats)
                else:
                    dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
            theta_plus1 = count * minibatch + 1.0
            beta = (theta_plus1 - minibatch) / theta_plus1
            backend.update_async(stats, beta, A, B, D, D_new)
            token = backend.maxdiff_token(step & 1)
            # stop test of the PREVIOUS step, now that this one is enqueued
            if pending is not None:
                ptoken, pit, pD = pending
                if backend.read_maxdiff(ptoken) < tol:  # dictionary_learning.py:161-162
                    # step `step` ran speculatively on the converged dictionary: its codes are
                    # not scattered, A / B / D_new of it are dropped
                    return pit, pD, x_local
            backend.scatter(x_stage, idx, n, x_local)
            pending = (token, it, D_new)
            D, D_new = D_new, D                 # the old D is the scratch of the next step
            count += 1
            step += 1
    if pending is not None:
        ptoken, pit, pD = pending
        if backend.read_maxdiff(ptoken) < tol:
            return pit, pD, x_local
    return maxiter, D, x_local


def dictionary_learning_sharded(y_local, D, alpha, x_local=None, tol=1.0e-3, minibatch=None, maxiter=1000,
                                lasso_method='cd', lasso_iter=10, lasso_tol=1.0e-5, random_seed=None,
                                group=None):
    """``decomp.dictionary_learning.solve(method='block_cd')`` for a sample-sharded problem.

    Every rank passes ITS rows of y (and of x), the same D, alpha, ``minibatch`` (the GLOBAL
    minibatch size) and ``random_seed``; ranks hold consecutive row ranges in rank order (sizes may
    differ).  Returns (it, D, x_local): it and D identical on all ranks, x_local this rank's codes
    in its original row order.  torch.distributed must be initialised (backend "nccl" = RCCL on
    ROCm) unless the world size is 1; one tiny all-gather of the row counts at set-up, then exactly
    one all-reduce of the [K, F+K] statistics per minibatch step."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from . import lasso
    if minibatch is None:
        raise NotImplementedError('Only online methods are implemented. minibatch is required.')
    lasso._dict_method_code(lasso_method)      # NotImplementedError for unknown solvers
    init = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if init else 1
    rank = dist.get_rank(group) if init else 0
    kind = _arrays.get_array_module(D)
    yd = _arrays.to_device(y_local)
    Dd = _arrays.to_device(D, copy=True)
    if x_local is None:
        xd = torch.ones((yd.shape[0], Dd.shape[0]), dtype=Dd.dtype, device=Dd.device)   # :58-59
    else:
        xd = _arrays.to_device(x_local, copy=True)
    counts = [int(yd.shape[0])]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, int(yd.shape[0]), group=group)
        counts = [int(c) for c in gathered]
    row0 = int(sum(counts[:rank]))
    n_total = int(sum(counts))
    _arrays.l2_normalize_(Dd, strict=True)
    backend = HipDictBackend(Dd, lasso_method, lasso_iter, lasso_tol, alpha)
    rng = np.random.RandomState(random_seed)

    def index_to_device(idx):
        return torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(Dd.device)
    it, Dout, xout = dict_loop(
        backend, yd, xd, row0, n_total, Dd, tol, minibatch, maxiter, rng, torch.empty_like,
        lambda shape: torch.zeros(shape, dtype=Dd.dtype, device=Dd.device), index_to_device,
        group=group, world_size=world,
        allreduce=comm_allreduce_ if (world > 1 and attach_communicator(Dd, group)) else None)
    return it, _arrays.to_caller(Dout, kind), _arrays.to_caller(xout, kind)
