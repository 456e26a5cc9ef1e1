"""n_samples-sharded NMF multiplicative update: one process per GPU, rows of y / x / mask
partitioned across ranks, D replicated.

Per iteration each rank runs the x update and its share of the D-side sums on its own
rows (``dcp_nmf_mu_stats_*``), the [K, F+K] (or [K, 2F]) statistics are summed over
ranks with ONE all-reduce (RCCL over xGMI when the process group's backend is "nccl"),
and every rank applies the identical D update (``dcp_nmf_mu_update_*``).  Nothing else
is communicated; the stop test runs redundantly and identically on every rank.

The reference has no multi-device path; the single-process semantics reproduced here are
those of decomp/nmf_methods/batch_mu.py:8-26.
"""
import ctypes

from . import _arrays, _hip


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

    def local_stats(self, D):
        lib, h = _arrays.lib_handle(D)
        fn = getattr(lib, 'dcp_nmf_mu_stats_' + self.sfx)
        _hip.check(h, fn(h, _arrays.ptr(self.y), _arrays.ptr(self.mask), _arrays.ptr(self.x),
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
    backend = HipStepBackend(y, m, x, Dd, lik)
    it, Dout = mu_loop(backend, Dd, tol, maxiter, group=group, world_size=world,
                       new_like=torch.empty_like)
    return it, Dout, backend.x
