"""MU iteration time at ragged shapes (atoms / rows / channels not multiples of the tiles): python tools/ragged_nmf.py"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from decomp_amd import _arrays, _hip
lib = _hip.load()
for (N, F, K) in [(65536, 4096, 256), (65536, 4096, 250), (65536, 4096, 252), (65000, 4000, 256), (65536, 4096, 64),
                  (65536, 4096, 50), (65536, 4096, 32), (65536, 4096, 30), (65536, 4096, 20), (65536, 4096, 10), (65536, 4096, 8), (65536, 4090, 256)]:
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    Y = torch.rand((N, F), generator=g, device='cuda')
    D = torch.rand((K, F), generator=g, device='cuda') + 0.1
    _arrays.l2_normalize_(D, strict=True)
    x = torch.ones((N, K), device='cuda')
    _, h = _arrays.lib_handle(D)
    it = ctypes.c_int(0)
    def solve(n):
        _hip.check(h, lib.dcp_nmf_mu_f32(h, _arrays.ptr(Y), None, _arrays.ptr(x), _arrays.ptr(D), N, F, K, 0,
                                         ctypes.c_float(0.0), n + 1, ctypes.byref(it), None, None), 'nmf')
    solve(3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); solve(10); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    W = 4.0 * N * K * F + 4.0 * N * K * K + 4.0 * K * K * F
    print('%6d x %4d k=%3d: %.3f ms/iteration  %.1f TF' % (N, F, K, ms, W / ms / 1e9))
    del Y, D, x
