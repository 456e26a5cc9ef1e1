"""``decomp.math_utils.linalg`` (linalg.py:9-38): batched inverse -- drop-in for ``inv``."""
from .. import _arrays, _hip


def inv(x):
    """Batch version of ``np.linalg.inv`` (linalg.py:9-38): ``x`` is [..., n, n] (any number of leading batch
    axes; the reference's CuPy branch handles up to three), float32 / float64 / complex64 / complex128.
    Computed by ``dcp_inv_*``: Gauss-Jordan with partial pivoting in double precision, one workgroup per
    matrix.  NumPy in -> NumPy out, torch CUDA in -> torch CUDA out.  A singular matrix gives inf / nan
    entries where NumPy raises LinAlgError."""
    import torch
    kind = _arrays.get_array_module(x)
    t = _arrays.to_device(x)
    if t.dim() < 2 or t.shape[-1] != t.shape[-2]:
        raise ValueError('Last 2 dimensions of the array must be square')
    n = t.shape[-1]
    batch = 1
    for s in t.shape[:-2]:
        batch *= s
    out = torch.empty_like(t)
    if batch and n:
        lib, h = _arrays.lib_handle(t)
        fn = getattr(lib, 'dcp_inv_' + _arrays.suffix(t))
        _hip.check(h, fn(h, _arrays.ptr(t), batch, n, _arrays.ptr(out)), 'dcp_inv')
    return _arrays.to_caller(out, kind)
