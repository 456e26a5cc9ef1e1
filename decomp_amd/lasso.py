"""Batched LASSO / NNLS -- drop-in for ``decomp.lasso`` on MI355X.

Same entry points, argument meaning, return convention and error behaviour as the
reference's decomp/lasso.py:19-189.  Everything after validation (``solve_fastpath``:
row-normalising A, the ista / acc_ista / fista / cd / parallel_cd / admm iterations,
masks, complex data) runs in libdecomp_hip.so: see include/decomp_hip.h ``dcp_lasso_*``
and decomp_amd/csrc/lasso_impl.hpp, lasso_extra.hpp.

'parallel_cd' consumes a host RNG stream in the reference (``RandomState(0).shuffle`` of
the commit vector every iteration, lasso.py:463,481): this module generates that stream
and hands it to the library as a table (``dcp_lasso_pcd_*``).
'admm' keeps the problem dtype; the reference silently promotes float32 / complex64
problems to double there (its ``rho * eye(K)`` is float64, lasso.py:603) -- same
iteration, the result agrees to the input precision.
"""
import ctypes

import numpy as np

from . import _arrays, _hip
from ._arrays import get_array_module
from .utils import assertion, dtype as _dtype

AVAILABLE_METHODS = ['ista', 'cd', 'acc_ista', 'fista', 'parallel_cd', 'admm']
AVAILABLE_NNLS_METHODS = ['ista_pos', 'cd_pos', 'acc_ista_pos', 'fista_pos',
                          'parallel_cd_pos', 'admm_pos']
_JITTER = 1.0e-15
_METHOD_CODE = {'ista': _hip.LASSO_ISTA, 'acc_ista': _hip.LASSO_ACC_ISTA,
                'fista': _hip.LASSO_FISTA, 'cd': _hip.LASSO_CD,
                'parallel_cd': _hip.LASSO_PARALLEL_CD, 'admm': _hip.LASSO_ADMM}


def _dict_method_code(lasso_method):
    """Code of a solver name for the dcp_dict_* entry points ('_pos' -> DCP_LASSO_POSITIVE)."""
    base = lasso_method[:-4] if lasso_method.endswith('_pos') else lasso_method
    if base not in _METHOD_CODE:                                      # lasso.py:157-159
        raise NotImplementedError('Method ' + base + ' is not yet implemented.')
    code = _METHOD_CODE[base]
    if lasso_method.endswith('_pos'):
        code |= _hip.LASSO_POSITIVE
    return code


def _dict_pcd_table(lasso_method, K, lasso_iter, like):
    """parallel_cd as the inner solver of the dictionary step: upload the shuffle table once and
    hand it to the library (dcp_dict_set_pcd_order).  Returns the device table (keep it alive for
    as long as the step entry points are called) or None for the other solvers."""
    import torch
    if not lasso_method.startswith('parallel_cd'):
        return None
    table = torch.from_numpy(_pcd_shuffle_table(K, max(int(lasso_iter), 1))).to(like.device)
    lib, h = _arrays.lib_handle(like)
    _hip.check(h, lib.dcp_dict_set_pcd_order(h, _arrays.ptr(table), table.shape[0], K),
               'dcp_dict_set_pcd_order')
    return table


def _pcd_shuffle_table(K, rows):
    """The reference's RNG stream for parallel_cd (lasso.py:463,481): it shuffles ONE 0/1
    vector cumulatively with RandomState(0); a shuffle's swaps do not depend on the
    content, so row i of this table (arange(K) after i + 1 shuffles) reproduces the
    vector of iteration i as ``table[i] < p`` for whatever p the device finds."""
    rng = np.random.RandomState(0)
    idx = np.arange(K, dtype=np.int32)
    table = np.empty((rows, K), dtype=np.int32)
    for i in range(rows):
        rng.shuffle(idx)
        table[i] = idx
    return table


class _ZerosLike(object):
    """Stand-in for the default x = zeros during validation."""
    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), dtype


def solve(y, A, alpha, x=None, tol=1.0e-3, method='ista', maxiter=1000,
          mask=None, **kwargs):
    """
    Solve  argmin_x {1 / (2 n) |y - xA|^2 + alpha |x|}  for every row of y.

    y: [..., n_channels], x: [..., n_features], A: [n_features, n_channels]; float or
    complex, all of one dtype; mask: float, y's shape or [n_channels].
    method: 'ista' | 'acc_ista' | 'fista' | 'cd' | 'parallel_cd' | 'admm'
    (+ '_pos' for non-negative x).
    Returns (it, x) as the reference does.
    """
    kind = get_array_module(y, A, x, mask)                            # lasso.py:71
    x_given = x
    if x is None:                                                     # lasso.py:73-74
        x = _ZerosLike(tuple(y.shape[:-1]) + (A.shape[0],), _arrays.np_dtype(y))

    assertion.assert_dtypes(y=y, A=A, x=x)                            # lasso.py:76-87
    assertion.assert_dtypes(mask=mask, dtypes='f')
    assertion.assert_nonnegative_host_or_device(mask)
    assertion.assert_ndim('A', A, ndim=2)
    assertion.assert_shapes('x', x, 'A', A, axes=1)
    assertion.assert_shapes('y', y, 'x', x, axes=np.arange(len(x.shape) - 1).tolist())
    assertion.assert_shapes('y', y, 'A', A, axes=[-1])
    if mask is not None and len(mask.shape) == 1:
        assertion.assert_shapes('y', y, 'mask', mask, axes=[-1])
    else:
        assertion.assert_shapes('y', y, 'mask', mask)
    if method not in AVAILABLE_METHODS + AVAILABLE_NNLS_METHODS:      # lasso.py:88-90
        raise ValueError('Available methods are {0:s}. Given {1:s}'.format(
                            str(AVAILABLE_METHODS), method))
    assert _arrays.np_dtype(A).kind != 'c' or method[-4:] != '_pos'   # lasso.py:92
    return solve_fastpath(y, A, alpha, x_given if x_given is not None else x, tol, maxiter,
                          method, kind, mask=mask, **kwargs)


def solve_fastpath(y, A, alpha, x, tol, maxiter, method, xp, mask=None, **kwargs):
    """lasso.py:97-189 on the GPU: no validation, no defaults (``x`` may be the
    zeros placeholder created by ``solve``).  ``xp`` is accepted for signature
    compatibility; the array kind is taken from ``y``."""
    import torch
    positive = False
    if method[-4:] == '_pos':
        method = method[:-4]
        positive = True
    if method not in _METHOD_CODE:                                    # lasso.py:157-159
        raise NotImplementedError('Method ' + method + ' is not yet implemented.')
    rho = 1.0
    if method == 'admm':                                              # lasso.py:155 (**kwargs)
        rho = float(kwargs.pop('rho', 1.0))
    if kwargs:
        raise TypeError('solve_fastpath() got an unexpected keyword argument %r'
                        % sorted(kwargs)[0])
    kind = 'torch' if _arrays.is_torch(y) else 'numpy'
    yd = _arrays.to_device(y)
    dev = yd.device.index
    Ad = _arrays.to_device(A, dev)
    F = Ad.shape[1]
    K = Ad.shape[0]
    batch_shape = tuple(yd.shape[:-1])
    y2 = yd.reshape(-1, F)
    N = y2.shape[0]
    if isinstance(x, _ZerosLike):
        xd = torch.zeros((N, K), dtype=yd.dtype, device=yd.device)
    else:
        xd = _arrays.to_device(x, dev, copy=True).reshape(N, K)
    sfx = _arrays.suffix(yd)
    rdt = {'f32': torch.float32, 'f64': torch.float64, 'c64': torch.float32,
           'c128': torch.float64}[sfx]
    mask_ndim = 0
    md = None
    if mask is not None:
        md = _arrays.to_device(mask, dev)
        if md.dtype != rdt:
            md = md.to(rdt)
        if md.dim() == 1:
            mask_ndim = 1
        else:
            mask_ndim = 2
            md = md.reshape(N, F)
        md = md.contiguous()
    lib, h = _arrays.lib_handle(yd)
    it = ctypes.c_int(0)
    y2 = y2.contiguous()
    if method == 'parallel_cd':
        table = torch.from_numpy(_pcd_shuffle_table(K, max(int(maxiter), 1))).to(yd.device)
        name = 'dcp_lasso_pcd_' + sfx
        rc = getattr(lib, name)(h, _arrays.ptr(y2), _arrays.ptr(md), mask_ndim, _arrays.ptr(Ad),
                                _arrays.ptr(xd), N, F, K, float(alpha), float(tol), int(maxiter),
                                1 if positive else 0, _arrays.ptr(table), table.shape[0],
                                ctypes.byref(it))
        if rc == _hip.ERR_REF_TYPEERROR:                              # lasso.py:509
            raise TypeError("_solve_cd_mask() missing 1 required positional argument: 'xp'")
    elif method == 'admm':
        name = 'dcp_lasso_admm_' + sfx
        rc = getattr(lib, name)(h, _arrays.ptr(y2), _arrays.ptr(md), mask_ndim, _arrays.ptr(Ad),
                                _arrays.ptr(xd), N, F, K, float(alpha), float(tol), int(maxiter),
                                1 if positive else 0, rho, ctypes.byref(it))
    else:
        name = 'dcp_lasso_' + sfx
        rc = getattr(lib, name)(h, _arrays.ptr(y2), _arrays.ptr(md), mask_ndim, _arrays.ptr(Ad),
                                _arrays.ptr(xd), N, F, K, float(alpha), float(tol), int(maxiter),
                                _METHOD_CODE[method], 1 if positive else 0, ctypes.byref(it))
    _hip.check(h, rc, name)
    out = xd.reshape(batch_shape + (K,))
    return it.value, _arrays.to_caller(out, kind)


# ---- proximal operators of the reference's public surface (lasso.py:192-241) -------------
# They are host utilities there (tests/test_lasso.py:15-56 call them on small arrays);
# the solvers above never call them: the same formulas live in the GEMM epilogues.
def soft_threshold_float(x, y, xp=np):
    return np.maximum(np.abs(x) - y, 0.0) * np.sign(x)


def soft_threshold_complex(x, y, xp=np):
    abs_x = np.abs(x)
    return np.maximum(abs_x - y, 0.0) * (x / (abs_x + _JITTER))


def soft_threshold_positive(x, y, xp=np):
    return np.maximum(x - y, 0.0)
