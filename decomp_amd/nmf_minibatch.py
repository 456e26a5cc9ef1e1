"""Stochastic (minibatch) multiplicative-update NMF on the GPU.

Host loops of the reference's decomp/nmf_methods/serizel.py:36-165 ('asg-mu', 'gsg-mu',
'asag-mu', 'gsag-mu') and decomp/nmf_methods/kasai.py:36-88 ('svrmu', 'svrmu-acc') around
the same gradient kernels as the full-batch path: per minibatch ONE library call updates
x in place and returns the two parts of the D gradient (``dcp_nmf_grads_*``); gradient
averaging, the D rule + l2_strict + max|dD| are ``dcp_axpby_*`` / ``dcp_nmf_apply_*``.
Epoch shuffles use the caller's ``np.random.RandomState`` exactly as the reference does.
"""
import ctypes

import numpy as np

from . import _arrays, _hip


class _Kernels(object):
    def __init__(self, D, lik):
        self.sfx = _arrays.suffix(D)
        self.lik = lik
        self.K, self.F = D.shape
        self.lib, _ = _arrays.lib_handle(D)
        self.md = ctypes.c_double(0.0)

    def grads(self, y_mb, m_mb, x_mb, D, n_x_updates, gpos, gneg):
        lib, h = _arrays.lib_handle(D)
        fn = getattr(lib, 'dcp_nmf_grads_' + self.sfx)
        _hip.check(h, fn(h, _arrays.ptr(y_mb), _arrays.ptr(m_mb), _arrays.ptr(x_mb), _arrays.ptr(D),
                         y_mb.shape[0], self.F, self.K, self.lik, int(n_x_updates),
                         _arrays.ptr(gpos), _arrays.ptr(gneg)), 'dcp_nmf_grads')

    def apply(self, D, P, Q, D_new, alpha=-1.0):
        lib, h = _arrays.lib_handle(D)
        fn = getattr(lib, 'dcp_nmf_apply_' + self.sfx)
        _hip.check(h, fn(h, _arrays.ptr(D), _arrays.ptr(P), _arrays.ptr(Q), float(alpha),
                         _arrays.ptr(D_new), self.K, self.F, ctypes.byref(self.md)), 'dcp_nmf_apply')
        return self.md.value

    def axpby(self, a, x, b, y):
        """y = a x + b y"""
        lib, h = _arrays.lib_handle(y)
        fn = getattr(lib, 'dcp_axpby_' + self.sfx)
        _hip.check(h, fn(h, y.numel(), float(a), _arrays.ptr(x), float(b), _arrays.ptr(y)), 'dcp_axpby')


class _UserKernels(_Kernels):
    """The same three operations with a user-supplied Likelihood (grads.py:12-13) standing in for
    the fused gradient kernel: the plugin's ``grad_x`` / ``grad_d`` run on arrays of the caller's
    kind, the update rule (``dcp_mu_quotient_*``) and everything else on the GPU."""

    def __init__(self, D, lik, kind):
        _Kernels.__init__(self, D, -1)
        self.user, self.kind = lik, kind

    def grads(self, y_mb, m_mb, x_mb, D, n_x_updates, gpos, gneg):
        from .nmf_methods.grads import mu_quotient

        def user(t):
            return None if t is None else _arrays.to_caller(t, self.kind)
        dev = D.device.index
        yu, mu, Du = user(y_mb), user(m_mb), user(D)
        for _ in range(int(n_x_updates)):                  # serizel.py:46-49
            pos, neg = self.user.grad_x(yu, user(x_mb), Du, mu)
            x_mb.copy_(_arrays.to_device(mu_quotient(user(x_mb), pos, neg), dev))
        pos, neg = self.user.grad_d(yu, user(x_mb), Du, mu)
        gpos.copy_(_arrays.to_device(pos, dev).expand(gpos.shape))
        gneg.copy_(_arrays.to_device(neg, dev).expand(gneg.shape))


def _kernels_for(D, lik, kind):
    return _Kernels(D, lik) if isinstance(lik, int) else _UserKernels(D, lik, kind)


def solve_serizel(y, D, x, tol, minibatch, maxiter, method, lik, mask, rng, forget_rate=0.5,
                  kind='torch'):
    """serizel.py:9-165.  y, x, mask: decomp_amd.utils.data.MinibatchData / NoneIterator.
    QUIRK kept: 'gsg-mu' runs the asg algorithm (serizel.py:23-25); on convergence the OLD
    D is returned (serizel.py:58-59)."""
    import torch
    kern = _kernels_for(D, lik, kind)
    gpos, gneg = torch.empty_like(D), torch.empty_like(D)
    D_new = torch.empty_like(D)
    averaged = method in ('asag-mu', 'gsag-mu')
    per_minibatch = method in ('asg-mu', 'gsg-mu', 'asag-mu')
    index = np.arange(y.size)
    for it in range(1, maxiter):
        rng.shuffle(index)
        y.shuffle(index)
        x.shuffle(index)
        mask.shuffle(index)
        if averaged:
            spos, sneg = torch.zeros_like(D), torch.zeros_like(D)
        for y_mb, x_mb, m_mb in zip(y, x, mask):
            kern.grads(y_mb, m_mb, x_mb, D, 1, gpos, gneg)
            P, Q = gpos, gneg
            if averaged:                                   # serizel.py:95-96
                kern.axpby(forget_rate, gpos, 1.0 - forget_rate, spos)
                kern.axpby(forget_rate, gneg, 1.0 - forget_rate, sneg)
                P, Q = spos, sneg
            if per_minibatch:
                if kern.apply(D, P, Q, D_new) < tol:
                    return it, D, x.array
                D, D_new = D_new, D
        if not per_minibatch:                              # gsag-mu: once per epoch
            if kern.apply(D, spos, sneg, D_new) < tol:
                return it, D, x.array
            D, D_new = D_new, D
    return maxiter, D, x.array


def solve_kasai(y, D, x, tol, minibatch, maxiter, method, lik, mask, rng, alpha=1.0, beta=0.5,
                kind='torch'):
    """kasai.py:10-88 (SVRMU / SVRMU-ACC)."""
    import torch
    if method == 'svrmu':
        iter_minibatch = 1
    else:                                                  # kasai.py:24-28
        F, K = D.shape
        N = x.shape[0]
        iter_minibatch = int(np.maximum(beta * F * (3 * K + 2 * N) / (3 * F * N + 2 * K), 1.0))
    kern = _kernels_for(D, lik, kind)
    index = np.arange(y.size)
    rng.shuffle(index)                                     # kasai.py:42-46: shuffled ONCE
    y.shuffle(index)
    x.shuffle(index)
    mask.shuffle(index)
    n_mb = y.n_loop
    prev_pos = torch.zeros((n_mb,) + tuple(D.shape), dtype=D.dtype, device=D.device)
    prev_neg = torch.zeros_like(prev_pos)
    full_pos, full_neg = torch.empty_like(D), torch.empty_like(D)
    gpos, gneg = torch.empty_like(D), torch.empty_like(D)
    P, Q = torch.empty_like(D), torch.empty_like(D)
    D_new = torch.empty_like(D)
    for it in range(1, maxiter):
        full_pos.zero_()
        full_neg.zero_()
        for y_mb, x_mb, m_mb in zip(y, x, mask):           # full gradient, kasai.py:53-60
            kern.grads(y_mb, m_mb, x_mb, D, 0, gpos, gneg)
            kern.axpby(1.0, gpos, 1.0, full_pos)
            kern.axpby(1.0, gneg, 1.0, full_neg)
        kern.axpby(0.0, full_pos, 1.0 / n_mb, full_pos)
        kern.axpby(0.0, full_neg, 1.0 / n_mb, full_neg)
        for k, (y_mb, x_mb, m_mb) in enumerate(zip(y, x, mask)):
            kern.grads(y_mb, m_mb, x_mb, D, iter_minibatch, gpos, gneg)
            # P = g+ + prev-[k] + full+ ; Q = g- + prev+[k] + full-       (kasai.py:74-75)
            kern.axpby(1.0, gpos, 0.0, P)
            kern.axpby(1.0, prev_neg[k], 1.0, P)
            kern.axpby(1.0, full_pos, 1.0, P)
            kern.axpby(1.0, gneg, 0.0, Q)
            kern.axpby(1.0, prev_pos[k], 1.0, Q)
            kern.axpby(1.0, full_neg, 1.0, Q)
            if kern.apply(D, P, Q, D_new, alpha=alpha) < tol:
                return it, D, x.array
            D, D_new = D_new, D
            kern.axpby(1.0, gpos, 0.0, prev_pos[k])
            kern.axpby(1.0, gneg, 0.0, prev_neg[k])
    return maxiter, D, x.array
