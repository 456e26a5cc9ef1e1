"""Online dictionary learning with lasso regularisation -- drop-in for
``decomp.dictionary_learning`` on MI355X.

Same entry point, argument meaning, return convention and error behaviour as the
reference's decomp/dictionary_learning.py:12-168 (``method='block_cd'``, Mairal et al.).
The epoch / minibatch loop and the seeded shuffle stay in Python (host logic, as in the
reference: the permutation comes from ``np.random.RandomState(random_seed)`` so the
minibatch composition is identical); each minibatch step -- LASSO inner solve, the
A / B statistics, the sequential atom sweep and max|D - D_new| -- is one call into
libdecomp_hip.so (``dcp_dict_step_*``, decomp_amd/csrc/dict_impl.hpp).

``mask`` selects the masked variant (solve_cd_mask, dictionary_learning.py:171-231): same
structure with a per-channel [K, F, K] Gram statistic (``dcp_dict_mask_step_*``; a parity
path, single GPU, memory K*F*K elements as in the reference).
"""
import ctypes

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
