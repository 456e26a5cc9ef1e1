"""Likelihoods of the multiplicative update -- drop-in for ``decomp.nmf_methods.grads``.

The reference's one genuine extension point (grads.py:7-93): ``nmf.solve(likelihood=obj)``
accepts any ``Likelihood`` instance whose ``grad_x`` / ``grad_d`` return the (positive,
negative) gradient parts.  ``nmf.solve`` runs the two built-in likelihoods (``Gaussian``, ``Poisson``) as
fused HIP kernels (``dcp_nmf_mu_*``, csrc/nmf_impl.hpp); their ``grad_x`` / ``grad_d`` / ``logp`` are
callable on their own as in the reference (``dcp_nmf_grad_x_*``, ``dcp_nmf_grads_*``,
``dcp_nmf_gauss_logp_*``), so a subclass that overrides one method can reach the others through
``super()``.  A subclass that overrides a method of the MU loop runs through a host loop
(decomp_amd.nmf._run_mu_user) in which the plugin
computes its gradient parts on arrays of the caller's kind (NumPy in -> NumPy arrays, torch
CUDA in -> torch CUDA tensors) and the library applies the reference's update rule
``cur * max(pos, 0) / max(neg, 1e-15)`` (``dcp_mu_quotient_*``), ``l2_strict`` and the
``max|D - D_new|`` stop test (``dcp_nmf_apply_*``) on the GPU.
"""
from .. import _arrays, _hip

_JITTER = 1.0e-15


def mu_quotient(cur, pos, neg):
    """cur * max(pos, 0) / max(neg, 1e-15) (grads.py:84,93) on the GPU.  ``pos`` / ``neg`` may be
    broadcastable to ``cur`` (the reference's Poisson parts are [1, K] / [K, 1]).  Arrays of
    either kind; the result has the kind of ``cur``."""
    kind = _arrays.get_array_module(cur)
    c = _arrays.to_device(cur)
    dev = c.device.index
    p = _arrays.to_device(pos, dev).to(c.dtype).expand(c.shape).contiguous()
    n = _arrays.to_device(neg, dev).to(c.dtype).expand(c.shape).contiguous()
    import torch
    out = torch.empty_like(c)
    rows = c.shape[0] if c.dim() > 1 else 1
    cols = c.numel() // max(rows, 1)
    lib, h = _arrays.lib_handle(c)
    fn = getattr(lib, 'dcp_mu_quotient_' + _arrays.suffix(c))
    _hip.check(h, fn(h, _arrays.ptr(c), _arrays.ptr(p), _arrays.ptr(n), rows, cols, _arrays.ptr(out)),
               'dcp_mu_quotient')
    return _arrays.to_caller(out, kind)


class Likelihood(object):
    """Base class for nmf likelihoods (grads.py:17-93).  Subclass it, implement ``grad_x`` and
    ``grad_d`` (each returns ``(grad_pos, grad_neg)``) and pass an instance as
    ``nmf.solve(..., likelihood=obj)``."""
    def __init__(self):
        pass

    def grad_x(self, y, x, d, mask):
        raise NotImplementedError

    def grad_d(self, y, x, d, mask):
        raise NotImplementedError

    def logp(self, y, x, d, mask):
        """ evaluate log likelihood """
        raise NotImplementedError

    def update_x(self, y, x, d, mask):
        """Multiplicative update rule for x (grads.py:77-84).  Returns the new x."""
        grad_pos, grad_neg = self.grad_x(y, x, d, mask)
        return mu_quotient(x, grad_pos, grad_neg)

    def update_d(self, y, x, d, mask):
        """Multiplicative update rule for d (grads.py:86-93).  Returns the new d."""
        grad_pos, grad_neg = self.grad_d(y, x, d, mask)
        return mu_quotient(d, grad_pos, grad_neg)


def _device_args(y, x, d, mask):
    """(kind, y, x, d, mask as contiguous device tensors of one dtype)."""
    kind = _arrays.get_array_module(y, x, d, mask)
    yd = _arrays.to_device(y)
    dev = yd.device.index
    xd, dd = _arrays.to_device(x, dev), _arrays.to_device(d, dev)
    md = _arrays.to_device(mask, dev)
    if md is not None and md.dtype != yd.dtype:
        md = md.to(yd.dtype)
    return kind, yd, xd, dd, md


def _grad_x(code, y, x, d, mask):
    """The two parts of the x gradient, [N, K] each (``dcp_nmf_grad_x_*``)."""
    import torch
    kind, yd, xd, dd, md = _device_args(y, x, d, mask)
    N, F = yd.shape
    K = dd.shape[0]
    pos = torch.empty((N, K), dtype=yd.dtype, device=yd.device)
    neg = torch.empty((N, K), dtype=yd.dtype, device=yd.device)
    lib, h = _arrays.lib_handle(yd)
    fn = getattr(lib, 'dcp_nmf_grad_x_' + _arrays.suffix(yd))
    _hip.check(h, fn(h, _arrays.ptr(yd), _arrays.ptr(md), _arrays.ptr(xd), _arrays.ptr(dd), N, F, K, code,
                     _arrays.ptr(pos), _arrays.ptr(neg)), 'dcp_nmf_grad_x')
    return kind, pos, neg


def _grad_d(code, y, x, d, mask):
    """The two parts of the D gradient, [K, F] each (``dcp_nmf_grads_*`` without an x update)."""
    import torch
    kind, yd, xd, dd, md = _device_args(y, x, d, mask)
    N, F = yd.shape
    K = dd.shape[0]
    pos = torch.empty((K, F), dtype=yd.dtype, device=yd.device)
    neg = torch.empty((K, F), dtype=yd.dtype, device=yd.device)
    lib, h = _arrays.lib_handle(yd)
    fn = getattr(lib, 'dcp_nmf_grads_' + _arrays.suffix(yd))
    _hip.check(h, fn(h, _arrays.ptr(yd), _arrays.ptr(md), _arrays.ptr(xd), _arrays.ptr(dd), N, F, K, code, 0,
                     _arrays.ptr(pos), _arrays.ptr(neg)), 'dcp_nmf_grads')
    return kind, pos, neg


class Gaussian(Likelihood):
    """Square loss (grads.py:96-135).  ``nmf.solve`` runs it as fused kernels (csrc/nmf_impl.hpp,
    DCP_LIK_L2); the methods below expose the same gradient parts through the reference's plugin surface, so
    that a subclass overriding one method can call the others via ``super()``.  Without a mask the negative
    parts use the Gram identities (x D) D^T = x (D D^T), x^T (x D) = (x^T x) D (rounding-level deviation)."""
    _code = _hip.LIK_L2

    def __init__(self, scale=1.0):
        self.scale = scale

    def grad_x(self, y, x, d, mask):
        """grads.py:108-115 -> (y.dot(d.T), f.dot(d.T)), [N, K] each."""
        kind, pos, neg = _grad_x(self._code, y, x, d, mask)
        return _arrays.to_caller(pos, kind), _arrays.to_caller(neg, kind)

    def grad_d(self, y, x, d, mask):
        """grads.py:117-125 -> (x.T.dot(y), x.T.dot(f)), [K, F] each."""
        kind, pos, neg = _grad_d(self._code, y, x, d, mask)
        return _arrays.to_caller(pos, kind), _arrays.to_caller(neg, kind)

    def logp(self, y, x, d, mask):
        """grads.py:127-135: sum((-0.5 ((y - x d) / scale)^2 - log(scale) - pi / 2) [* mask]), a scalar of
        y's dtype (``dcp_nmf_gauss_logp_*``, accumulated in double precision)."""
        import ctypes
        kind, yd, xd, dd, md = _device_args(y, x, d, mask)
        N, F = yd.shape
        K = dd.shape[0]
        out = ctypes.c_double(0.0)
        lib, h = _arrays.lib_handle(yd)
        fn = getattr(lib, 'dcp_nmf_gauss_logp_' + _arrays.suffix(yd))
        _hip.check(h, fn(h, _arrays.ptr(yd), _arrays.ptr(md), _arrays.ptr(xd), _arrays.ptr(dd), N, F, K,
                         float(self.scale), ctypes.byref(out)), 'dcp_nmf_gauss_logp')
        if kind == 'torch':
            import torch
            return torch.tensor(out.value, dtype=yd.dtype, device=yd.device)
        return _arrays.np_dtype(yd).type(out.value)


class Poisson(Likelihood):
    """KL loss (grads.py:138-160).  Fused in ``nmf.solve`` (DCP_LIK_KL); gradient parts exposed as for
    ``Gaussian``, with the reference's shapes: without a mask the negative parts are the broadcastable
    [1, K] / [K, 1] sums (grads.py:146, 155)."""
    _code = _hip.LIK_KL

    def grad_x(self, y, x, d, mask):
        """grads.py:143-150."""
        kind, pos, neg = _grad_x(self._code, y, x, d, mask)
        if mask is None:
            neg = neg[:1].contiguous()          # d.T.sum(axis=0, keepdims=True)
        return _arrays.to_caller(pos, kind), _arrays.to_caller(neg, kind)

    def grad_d(self, y, x, d, mask):
        """grads.py:152-160."""
        kind, pos, neg = _grad_d(self._code, y, x, d, mask)
        if mask is None:
            neg = neg[:, :1].contiguous()       # x.T.sum(axis=1, keepdims=True)
        return _arrays.to_caller(pos, kind), _arrays.to_caller(neg, kind)

    def logp(self, y, x, d, mask):
        """grads.py:162-170 reads ``self.scale``, which Poisson never sets: the reference raises
        AttributeError here, and so does this drop-in."""
        raise AttributeError("'Poisson' object has no attribute 'scale'")


def fused_code(likelihood):
    """The kernel code of a likelihood that ``nmf.solve`` may run as fused kernels: an instance of Gaussian /
    Poisson -- or of a subclass that overrides none of the four methods the MU loop calls (a subclass that only
    adds ``logp`` or bookkeeping keeps the fused path); None otherwise."""
    for base in (Gaussian, Poisson):
        if isinstance(likelihood, base):
            cls = type(likelihood)
            if all(getattr(cls, n) is getattr(base, n) for n in ('grad_x', 'grad_d', 'update_x', 'update_d')):
                return base._code
    return None


def get_likelihood(likelihood):
    """grads.py:7-14."""
    if likelihood in ['l2', 'gaussian']:
        return Gaussian()
    if likelihood in ['kl', 'poisson']:
        return Poisson()
    if isinstance(likelihood, Likelihood):
        return likelihood
    raise NotImplementedError('Likelihood {} is not implemented for nmf'.format(likelihood))
