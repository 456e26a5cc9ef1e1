"""Eigenvalue bounds -- drop-in for ``decomp.math_utils.eigen`` (eigen.py:9-20)."""
import ctypes  # noqa: F401

from .. import _arrays, _hip


def spectral_radius_Gershgorin(X, xp=None, keepdims=False):
    """An upper bound of the largest eigenvalue of the (symmetric) matrices X[..., n, n] by
    Gershgorin's circle theorem: max_j sum_i |X_ij|, shape [..., 1] (eigen.py:20; like the
    reference, ``keepdims`` is accepted and the trailing axis is always kept).
    Computed by ``dcp_gershgorin_*`` (one workgroup per matrix)."""
    import torch
    kind = _arrays.get_array_module(X)
    t = _arrays.to_device(X)
    if t.dim() < 2 or t.shape[-1] != t.shape[-2]:
        raise ValueError('X should be a matrix or a batch of matrices, shape [..., n, n]')
    n = t.shape[-1]
    batch_shape = tuple(t.shape[:-2])
    batch = 1
    for s in batch_shape:
        batch *= s
    rdt = {torch.complex64: torch.float32, torch.complex128: torch.float64}.get(t.dtype, t.dtype)
    out = torch.empty((batch,), dtype=rdt, device=t.device)
    if batch:
        lib, h = _arrays.lib_handle(t)
        fn = getattr(lib, 'dcp_gershgorin_' + _arrays.suffix(t))
        _hip.check(h, fn(h, _arrays.ptr(t), batch, n, _arrays.ptr(out)), 'dcp_gershgorin')
    # eigen.py:20 reduces axis -2, then axis -1 with keepdims=True: result shape [..., 1]
    return _arrays.to_caller(out.reshape(batch_shape + (1,)), kind)


def spectral_radius_svd(X, xp=None):
    """eigen.py:4-6.  Not on the hot path and not a kernel of this library."""
    raise NotImplementedError('spectral_radius_svd is not part of the MI355X hot path; '
                              'use spectral_radius_Gershgorin')
