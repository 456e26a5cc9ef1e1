"""Row normalisation -- drop-in for ``decomp.utils.normalize`` (decomp/utils/normalize.py:2-21).

Same names and argument order as the reference (``l2(U, xp, axis=-1)``,
``l2_strict(U, xp, axis=-1)``); ``xp`` is accepted for signature compatibility and ignored
(the array kind decides: NumPy in -> NumPy out, torch CUDA in -> torch CUDA out).  The
arithmetic is ``row_normalize_kernel`` of libdecomp_hip.so through ``dcp_l2_normalize_*``;
a new array is returned and the input is left untouched, as in the reference.
"""
from .. import _arrays


def _normalize(U, axis, strict):
    kind = _arrays.get_array_module(U)
    t = _arrays.to_device(U)
    nd = t.dim()
    if nd == 0:
        raise ValueError('normalize needs at least a 1-d array')
    ax = axis % nd
    moved = t.movedim(ax, -1) if ax != nd - 1 else t
    shape = moved.shape
    import torch
    # a fresh C-contiguous copy (the kernel works in place; clone() alone would keep the strides
    # of a transposed view)
    flat = moved.reshape(-1, shape[-1]).clone(memory_format=torch.contiguous_format)
    if flat.numel():
        _arrays.l2_normalize_(flat, strict=strict)
    out = flat.reshape(shape)
    if ax != nd - 1:
        out = out.movedim(-1, ax).contiguous()
    return _arrays.to_caller(out, kind)


def l2(U, xp=None, axis=-1):
    """U / sqrt(max(sum |U|^2, 1)) along ``axis`` (normalize.py:2-10)."""
    return _normalize(U, axis, strict=False)


def l2_strict(U, xp=None, axis=-1):
    """U / sqrt(sum |U|^2) along ``axis`` (normalize.py:13-21)."""
    return _normalize(U, axis, strict=True)
