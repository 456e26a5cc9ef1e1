"""Array plumbing between the caller's arrays and device memory.

The reference's array-module handle ``xp`` (decomp/utils/cp_compat.py:9-24) becomes:
NumPy arrays are copied to the current GPU and results come back as NumPy arrays;
torch CUDA tensors are used in place and results come back as torch tensors.  torch is
used ONLY for device memory, streams and (in decomp_amd.sharded) torch.distributed --
every arithmetic operation on the hot path is a kernel of libdecomp_hip.so.
"""
import ctypes

import numpy as np

try:  # torch is plumbing: device memory + streams
    import torch
except ImportError:  # pragma: no cover
    torch = None

from . import _hip

_SUFFIX = {np.dtype(np.float32): 'f32', np.dtype(np.float64): 'f64',
           np.dtype(np.complex64): 'c64', np.dtype(np.complex128): 'c128'}


def _torch_to_np_dtype(dt):
    return {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64),
            torch.complex64: np.dtype(np.complex64), torch.complex128: np.dtype(np.complex128),
            torch.float16: np.dtype(np.float16), torch.int32: np.dtype(np.int32),
            torch.int64: np.dtype(np.int64), torch.bool: np.dtype(np.bool_),
            torch.uint8: np.dtype(np.uint8), torch.int8: np.dtype(np.int8),
            torch.int16: np.dtype(np.int16)}.get(dt, np.dtype(np.void))


def is_torch(a):
    return torch is not None and isinstance(a, torch.Tensor)


def np_dtype(a):
    """NumPy dtype of a NumPy array or torch tensor."""
    if is_torch(a):
        return _torch_to_np_dtype(a.dtype)
    return np.dtype(a.dtype)


def suffix(a):
    dt = np_dtype(a)
    if dt not in _SUFFIX:
        raise TypeError('unsupported dtype %s' % dt)
    return _SUFFIX[dt]


def get_array_module(*arrays):
    """'numpy' or 'torch' -- the kind of the first array; every other non-None array
    must be of the same kind (cp_compat.py:9-15 raises the same TypeError)."""
    kind = 'torch' if is_torch(arrays[0]) else 'numpy'
    for a in arrays:
        if a is None:
            continue
        if ('torch' if is_torch(a) else 'numpy') != kind:
            raise TypeError('All the data types should be the same.')
    return kind


def current_device():
    if torch is None or not torch.cuda.is_available():
        raise _hip.HipLibraryError(
            'no HIP device is visible to torch: decomp_amd computes only on the GPU '
            '(there is no CPU fallback).')
    return torch.cuda.current_device()


def to_device(a, device=None, copy=False):
    """A contiguous torch CUDA tensor holding ``a`` (None stays None)."""
    if a is None:
        return None
    if is_torch(a):
        if not a.is_cuda:
            raise TypeError('torch tensors passed to decomp_amd must live on the GPU')
        t = a.contiguous()
        if copy and t.data_ptr() == a.data_ptr():
            t = t.clone()
        return t
    dev = current_device() if device is None else device
    arr = np.ascontiguousarray(a)
    return torch.from_numpy(arr).to('cuda:%d' % dev)


def to_caller(t, kind):
    """Back to the caller's array kind."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):     # streamed containers hand back host arrays
        return t
    if kind == 'torch':
        return t
    return t.cpu().numpy()


def ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def lib_handle(t):
    """(lib, handle) for the device ``t`` lives on, with the handle bound to torch's
    current stream on that device."""
    lib = _hip.load()
    dev = t.device.index
    h = _hip.handle(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _hip.check(h, lib.dcp_set_stream(h, ctypes.c_void_p(stream)), 'dcp_set_stream')
    return lib, h


def count_negative(t):
    """Number of elements failing ``x >= 0`` (HIP scan kernel)."""
    lib, h = lib_handle(t)
    out = ctypes.c_int64(0)
    fn = getattr(lib, 'dcp_count_negative_' + suffix(t))
    _hip.check(h, fn(h, ptr(t), t.numel(), ctypes.byref(out)), 'dcp_count_negative')
    return out.value


def l2_normalize_(t, strict):
    """In-place row normalisation of a [K, F] device array (normalize.py:2-21)."""
    lib, h = lib_handle(t)
    fn = getattr(lib, 'dcp_l2_normalize_' + suffix(t))
    _hip.check(h, fn(h, ptr(t), t.shape[0], t.shape[1], 1 if strict else 0),
               'dcp_l2_normalize')
    return t
