"""Non-negative matrix factorisation -- drop-in for ``decomp.nmf`` on MI355X.

Same entry point, argument meaning, return convention and error behaviour as the
reference's decomp/nmf.py:16-113: the full-batch multiplicative update
(``minibatch=None, method='mu'``) and the stochastic minibatch variants
(decomp_amd/nmf_minibatch.py).  The full-batch iteration itself
(decomp/nmf_methods/batch_mu.py:8-26 with the update rules of
decomp/nmf_methods/grads.py:77-160) runs in libdecomp_hip.so: see
include/decomp_hip.h ``dcp_nmf_mu_*`` and decomp_amd/csrc/nmf_impl.hpp.
"""
import ctypes

import numpy as np

from . import _arrays, _hip
from ._arrays import get_array_module
from .utils import assertion

BATCH_METHODS = ['mu']
MINIBATCH_METHODS = [
    'asg-mu', 'gsg-mu', 'asag-mu', 'gsag-mu',  # Serizel et al.
    'svrmu', 'svrmu-acc',                      # Kasai et al.
    ]
_JITTER = 1.0e-15


def _likelihood_code(likelihood):
    """grads.py:7-14: the kernel code (an int) of a built-in likelihood, or the user-supplied
    ``Likelihood`` instance itself (grads.py:12-13), which then runs through the host loop
    ``_run_mu_user`` / ``nmf_minibatch._UserKernels``."""
    from .nmf_methods import grads
    if isinstance(likelihood, str):
        if likelihood in ('l2', 'gaussian'):
            return _hip.LIK_L2
        if likelihood in ('kl', 'poisson'):
            return _hip.LIK_KL
    elif isinstance(likelihood, grads.Likelihood):
        code = grads.fused_code(likelihood)
        return likelihood if code is None else code
    raise NotImplementedError('Likelihood {} is not implemented for nmf'.format(likelihood))


class _OnesLike(object):
    """Stand-in for the default x = ones((N, K)) during validation, so that the default
    is materialised directly in device memory."""
    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), dtype


def solve(y, D, x=None, tol=1.0e-3, minibatch=None, maxiter=1000, method='mu',
          likelihood='l2', mask=None, random_seed=None, **kwargs):
    """
    Non-negative matrix factorisation  argmin_{x, D} |y - xD|^2,  x >= 0, D >= 0,
    |D_j| = 1, by multiplicative updates.

    y: [n_samples, n_channels], x: [n_samples, n_features], D: [n_features, n_channels],
    mask (optional): [n_samples, n_channels], 0 marks a missing entry; float32 or
    float64, all arrays of the same dtype.  NumPy arrays (copied to the GPU, results
    returned as NumPy) or torch CUDA tensors (results returned as torch tensors).
    Out of core (nmf.py:93-103): with a minibatch method, a torch CUDA ``D`` and NumPy
    ``y`` / ``x`` / ``mask``, the NumPy arrays stay in pinned host memory and are streamed
    through the GPU minibatch by minibatch (utils.data.AsyncMinibatchData); ``x`` then
    comes back as a NumPy array.

    Returns (it, D, x) exactly as the reference: ``it`` is the iteration at which
    max|D - D_new| < tol was met, or ``maxiter`` when it never was.
    """
    kind = get_array_module(D)
    x_given = x
    if x is None:                                                     # nmf.py:53-54
        x = _OnesLike((y.shape[0], D.shape[0]), _arrays.np_dtype(y))

    assertion.assert_dtypes(y=y, D=D, x=x)                            # nmf.py:56-63
    assertion.assert_dtypes(y=y, D=D, x=x, mask=mask, dtypes='f')
    assertion.assert_shapes('x', x, 'D', D, axes=1)
    assertion.assert_shapes('y', y, 'D', D, axes=[-1])
    assertion.assert_shapes('y', y, 'mask', mask)
    assertion.assert_ndim('y', y, 2)
    assertion.assert_ndim('D', D, 2)
    assertion.assert_ndim('x', x, 2)

    if minibatch is None or kind == 'numpy':
        get_array_module(D, x_given)
    # out of core (nmf.py:93-103): device D, host data, minibatch method
    streamed = (minibatch is not None and kind == 'torch' and
                any(a is not None and not _arrays.is_torch(a) for a in (y, x_given, mask)))

    # ---- from here on everything lives on the GPU ----
    import torch
    D_dev = _arrays.to_device(D, copy=True)           # normalised in place below
    dev = D_dev.device.index
    assertion.assert_nonnegative(D_dev)                               # nmf.py:64-65
    if streamed:
        return _solve_streamed(y, D_dev, x_given, tol, minibatch, maxiter, method, likelihood,
                               mask, random_seed, kwargs)
    if x_given is None:
        x_dev = torch.ones(x.shape, dtype=D_dev.dtype, device=D_dev.device)
    else:
        x_dev = _arrays.to_device(x_given, dev, copy=True)   # updated in place
    assertion.assert_nonnegative(x_dev)
    lik = None
    if isinstance(likelihood, str) and likelihood in ['kl']:          # nmf.py:67-68
        y_dev = _arrays.to_device(y, dev)
        assertion.assert_nonnegative(y_dev)
    else:
        y_dev = None

    _arrays.l2_normalize_(D_dev, strict=True)                         # nmf.py:70

    if minibatch is None:
        get_array_module(y, D, x_given)                               # nmf.py:75
        if method == 'mu':
            if kwargs:  # batch_mu.solve accepts no extra keyword (nmf.py:77, batch_mu.py:8)
                raise TypeError('solve() got an unexpected keyword argument %r'
                                % sorted(kwargs)[0])
            lik = _likelihood_code(likelihood)
            get_array_module(y, mask)
            if not isinstance(lik, int):
                it, D_dev, x_dev = _run_mu_user(y, mask, x_dev, D_dev, lik, tol, maxiter, kind)
                return it, _arrays.to_caller(D_dev, kind), _arrays.to_caller(x_dev, kind)
            if y_dev is None:
                y_dev = _arrays.to_device(y, dev)
            m_dev = _arrays.to_device(mask, dev)
            it = _run_mu(y_dev, m_dev, x_dev, D_dev, lik, tol, maxiter)
            return it, _arrays.to_caller(D_dev, kind), _arrays.to_caller(x_dev, kind)
        raise NotImplementedError('Batch-NMF with {} algorithm is not yet '
                                  'implemented.'.format(method))
    # ---- stochastic variants: minibatch containers on the GPU (nmf.py:82-111) ----
    from .utils.data import MinibatchData, NoneIterator
    from . import nmf_minibatch
    if method not in MINIBATCH_METHODS:
        raise NotImplementedError('NMF with {} algorithm is not yet '
                                  'implemented.'.format(method))
    get_array_module(y, D, x_given, mask)
    lik = _likelihood_code(likelihood)
    if y_dev is None:
        y_dev = _arrays.to_device(y, dev)
    ybat = MinibatchData(y_dev, minibatch)
    xbat = MinibatchData(x_dev, minibatch)
    mbat = NoneIterator() if mask is None else MinibatchData(_arrays.to_device(mask, dev), minibatch)
    rng = np.random.RandomState(random_seed)
    if method in ['asg-mu', 'gsg-mu', 'asag-mu', 'gsag-mu']:
        it, Dout, xout = nmf_minibatch.solve_serizel(ybat, D_dev, xbat, tol, minibatch, maxiter,
                                                     method, lik, mbat, rng, kind=kind, **kwargs)
    else:
        it, Dout, xout = nmf_minibatch.solve_kasai(ybat, D_dev, xbat, tol, minibatch, maxiter,
                                                   method, lik, mbat, rng, kind=kind, **kwargs)
    return it, _arrays.to_caller(Dout, kind), _arrays.to_caller(xout, kind)


def _solve_streamed(y, D_dev, x_given, tol, minibatch, maxiter, method, likelihood, mask,
                    random_seed, kwargs):
    """nmf.py:93-111: every array that is a NumPy array stays on the host and is streamed
    (x with write-back); device arrays use the in-core container.  Returns (it, D, x) with
    x a NumPy array when it was streamed."""
    import torch
    from .utils.data import MinibatchData, AsyncMinibatchData, NoneIterator
    from . import nmf_minibatch
    if method not in MINIBATCH_METHODS:
        raise NotImplementedError('NMF with {} algorithm is not yet '
                                  'implemented.'.format(method))
    dev = D_dev.device.index

    def dataset(a, needs_update):
        if a is None:
            return NoneIterator()
        if _arrays.is_torch(a):
            t = _arrays.to_device(a, dev, copy=needs_update)
            if needs_update:
                assertion.assert_nonnegative(t)                       # nmf.py:65
            return MinibatchData(t, minibatch)
        if needs_update or (isinstance(likelihood, str) and likelihood in ['kl']):
            assertion.assert_nonnegative_host_or_device(a)            # nmf.py:65,67-68
        return AsyncMinibatchData(a, minibatch, needs_update=needs_update, device=dev)

    if x_given is None:                                               # nmf.py:53-54: ones in D's module
        x_given = torch.ones((y.shape[0], D_dev.shape[0]), dtype=D_dev.dtype, device=D_dev.device)
    if _arrays.is_torch(y) and isinstance(likelihood, str) and likelihood in ['kl']:
        assertion.assert_nonnegative(_arrays.to_device(y, dev))
    _arrays.l2_normalize_(D_dev, strict=True)                         # nmf.py:70
    lik = _likelihood_code(likelihood)
    xbat = dataset(x_given, True)
    ybat = dataset(y, False)
    mbat = dataset(mask, False)
    rng = np.random.RandomState(random_seed)
    if method in ['asg-mu', 'gsg-mu', 'asag-mu', 'gsag-mu']:
        it, Dout, xout = nmf_minibatch.solve_serizel(ybat, D_dev, xbat, tol, minibatch, maxiter,
                                                     method, lik, mbat, rng, **kwargs)
    else:
        it, Dout, xout = nmf_minibatch.solve_kasai(ybat, D_dev, xbat, tol, minibatch, maxiter,
                                                   method, lik, mbat, rng, **kwargs)
    return it, Dout, xout


def _run_mu(y, mask, x, D, lik, tol, maxiter, resid_trace=None):
    """batch_mu.solve on device arrays; x and D are updated in place.  Returns it."""
    lib, h = _arrays.lib_handle(D)
    sfx = _arrays.suffix(D)
    N, F = y.shape
    K = D.shape[0]
    ctype = ctypes.c_float if sfx == 'f32' else ctypes.c_double
    it = ctypes.c_int(0)
    last = ctype(0)
    trace = None
    if resid_trace is not None:
        trace = (ctype * max(int(maxiter), 1))()
    fn = getattr(lib, 'dcp_nmf_mu_' + sfx)
    rc = fn(h, _arrays.ptr(y), _arrays.ptr(mask), _arrays.ptr(x), _arrays.ptr(D),
            N, F, K, lik, ctype(tol), int(maxiter), ctypes.byref(it), ctypes.byref(last),
            trace)
    _hip.check(h, rc, 'dcp_nmf_mu_' + sfx)
    if resid_trace is not None:
        n_done = it.value if it.value < maxiter else maxiter - 1
        resid_trace.extend(float(trace[i]) for i in range(max(n_done, 0)))
    return it.value


def _run_mu_user(y, mask, x_dev, D_dev, lik, tol, maxiter, kind):
    """batch_mu.py:8-26 with a user-supplied Likelihood (grads.py:12-13).  The plugin's
    ``update_x`` / ``update_d`` see arrays of the caller's kind (NumPy or torch CUDA); the
    inherited update rule, ``l2_strict`` and the stop test run on the GPU
    (``dcp_mu_quotient_*``, ``dcp_l2_normalize_diff_*``).  Returns (it, D, x) as device tensors."""
    import torch
    dev = D_dev.device.index
    sfx = _arrays.suffix(D_dev)
    K, F = D_dev.shape
    md = ctypes.c_double(0.0)

    def user(t):
        return _arrays.to_caller(t, kind)

    y_u = y if kind == 'numpy' else _arrays.to_device(y, dev)
    m_u = mask if (mask is None or kind == 'numpy') else _arrays.to_device(mask, dev)
    x, D = x_dev, D_dev
    D_new = torch.empty_like(D)
    for it in range(1, maxiter):                                       # batch_mu.py:16
        x = _arrays.to_device(lik.update_x(y_u, user(x), user(D), m_u), dev)
        U = _arrays.to_device(lik.update_d(y_u, user(x), user(D), m_u), dev)
        if U.shape != D.shape or U.dtype != D.dtype:
            raise ValueError('update_d must return an array like D: %s %s' % (tuple(U.shape), U.dtype))
        lib, h = _arrays.lib_handle(D)
        fn = getattr(lib, 'dcp_l2_normalize_diff_' + sfx)
        _hip.check(h, fn(h, _arrays.ptr(U), _arrays.ptr(D), _arrays.ptr(D_new), K, F, 1,
                         ctypes.byref(md)), 'dcp_l2_normalize_diff')
        if md.value < tol:                                             # batch_mu.py:22
            return it, D_new, x
        D, D_new = D_new, D
    return maxiter, D, x
