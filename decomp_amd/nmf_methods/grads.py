"""Likelihoods of the multiplicative update -- drop-in for ``decomp.nmf_methods.grads``.

The reference's one genuine extension point (grads.py:7-93): ``nmf.solve(likelihood=obj)``
accepts any ``Likelihood`` instance whose ``grad_x`` / ``grad_d`` return the (positive,
negative) gradient parts.  The two built-in likelihoods (``Gaussian``, ``Poisson``) are fused
HIP kernels (``dcp_nmf_mu_*``, csrc/nmf_impl.hpp) and are only *named* by these classes; a
user subclass runs through a host loop (decomp_amd.nmf._run_mu_user) in which the plugin
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


class Gaussian(Likelihood):
    """Square loss (grads.py:96-135).  Fused kernels: csrc/nmf_impl.hpp (DCP_LIK_L2)."""
    _code = _hip.LIK_L2

    def __init__(self, scale=1.0):
        self.scale = scale


class Poisson(Likelihood):
    """KL loss (grads.py:138-160).  Fused kernels: csrc/nmf_impl.hpp (DCP_LIK_KL)."""
    _code = _hip.LIK_KL


def get_likelihood(likelihood):
    """grads.py:7-14."""
    if likelihood in ['l2', 'gaussian']:
        return Gaussian()
    if likelihood in ['kl', 'poisson']:
        return Poisson()
    if isinstance(likelihood, Likelihood):
        return likelihood
    raise NotImplementedError('Likelihood {} is not implemented for nmf'.format(likelihood))
