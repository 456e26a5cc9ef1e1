"""Helpers shared by the -m gpu tests (which call the product through the C ABI and
check it against the CPU oracle)."""
import ctypes

import numpy as np


def torch_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def gemm_hip(form, A, B, ksplits=1, tile=0):
    """C = op(A).op(B) through dcp_gemm_* (form 0 NT, 1 NN, 2 TN)."""
    import torch
    from decomp_amd import _arrays, _hip
    a, b = torch_dev(A), torch_dev(B)
    if form == 0:
        M, K = A.shape; N = B.shape[0]
    elif form == 1:
        M, K = A.shape; N = B.shape[1]
    else:
        K, M = A.shape; N = B.shape[1]
    c = torch.empty((M, N), dtype=a.dtype, device='cuda')
    lib, h = _arrays.lib_handle(a)
    fn = getattr(lib, 'dcp_gemm_' + _arrays.suffix(a))
    _hip.check(h, fn(h, form, _arrays.ptr(a), _arrays.ptr(b), _arrays.ptr(c), M, N, K,
                     ksplits, tile), 'dcp_gemm')
    torch.cuda.synchronize()
    return c.cpu().numpy()


def gemm_ref(form, A, B):
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    if form == 0:
        return A64 @ B64.T
    if form == 1:
        return A64 @ B64
    return A64.T @ B64


def gemm_bound(form, A, B):
    """sum_k |a||b| : the scale fp32 rounding errors are relative to."""
    return gemm_ref(form, np.abs(A), np.abs(B))
