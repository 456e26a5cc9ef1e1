#!/usr/bin/env python3
"""float64 / complex128 paths (the reference's default dtype is float64): one NMF MU iteration and
a plain GEMM through the test hook, with the fp64 roofline (78.6 TFLOP/s matrix = vector on MI355X)
beside it.   python tools/bench_f64.py [--rows 16384]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
from decomp_amd import _arrays, _hip  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=16384)
    ap.add_argument('--f', type=int, default=4096)
    ap.add_argument('--k', type=int, default=256)
    a = ap.parse_args()
    N, F, K = a.rows, a.f, a.k
    g = torch.Generator(device='cuda')
    g.manual_seed(1)
    dd = torch.float64
    Dt = torch.randn((K, F), generator=g, device='cuda', dtype=dd).clamp_(min=0)
    xt = torch.randn((N, K), generator=g, device='cuda', dtype=dd).clamp_(min=0)
    Y = xt @ Dt + 0.1 * torch.randn((N, F), generator=g, device='cuda', dtype=dd).abs_()
    D0 = (Dt + 0.3 * torch.randn((K, F), generator=g, device='cuda', dtype=dd)).clamp_(min=0.1)
    lib, h = _arrays.lib_handle(Y)
    C = torch.empty((N, K), device='cuda', dtype=dd)
    for form, (A, B, M_, N_, K_) in {0: (Y, D0, N, K, F), 2: (xt, Y, K, F, N)}.items():
        out = torch.empty((M_, N_), device='cuda', dtype=dd)
        for splits in ((1,) if form == 0 else (16,)):
            for tile in (0, 3, 4, 5, 6, 2):
                ms = timed(lambda: _hip.check(h, lib.dcp_gemm_f64(h, form, _arrays.ptr(A), _arrays.ptr(B),
                                                                   _arrays.ptr(out), M_, N_, K_, splits, tile),
                                              'gemm'), 5)
                print('gemm f64 form %d  %dx%dx%d splits %2d tile %d: %.3f ms  %.1f TFLOP/s (%.0f %% of 78.6)'
                      % (form, M_, N_, K_, splits, tile, ms, 2.0 * M_ * N_ * K_ / ms / 1e9,
                         2.0 * M_ * N_ * K_ / ms / 1e9 / 78.6 * 100))
    D = D0.clone()
    _arrays.l2_normalize_(D, strict=True)
    x = torch.ones((N, K), device='cuda', dtype=dd)
    it = ctypes.c_int(0)

    def run(n):
        _hip.check(h, lib.dcp_nmf_mu_f64(h, _arrays.ptr(Y), None, _arrays.ptr(x), _arrays.ptr(D), N, F, K,
                                         0, ctypes.c_double(0.0), n + 1, ctypes.byref(it), None, None),
                   'nmf_mu_f64')
    run(2)
    ms = timed(lambda: run(10), 1) / 10
    W = 4.0 * N * K * F + 4.0 * N * K * K + 4.0 * K * K * F
    print('nmf mu f64 %dx%d k=%d: %.3f ms/iter  %.1f TFLOP/s algorithmic (%.0f %% of 78.6)'
          % (N, F, K, ms, W / ms / 1e9, W / ms / 1e9 / 78.6 * 100))


def complex128_lasso():
    """dictionary-learning-like LASSO call in complex128 (the reference's default complex dtype)."""
    import time
    import numpy as np
    from decomp_amd import lasso
    N, F, K = 8192, 2048, 256
    g = torch.Generator(device='cuda')
    g.manual_seed(3)

    def randn(*s):
        return torch.complex(torch.randn(s, generator=g, device='cuda', dtype=torch.float64),
                             torch.randn(s, generator=g, device='cuda', dtype=torch.float64))
    A = randn(K, F)
    xt = randn(N, K) * (torch.rand((N, K), generator=g, device='cuda') < 0.05)
    y = xt @ A + 0.1 * randn(N, F)
    lasso.solve(y, A, 0.05, tol=1e-12, method='ista', maxiter=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it, x = lasso.solve(y, A, 0.05, tol=1e-12, method='ista', maxiter=30)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flops = 8.0 * N * F * K + 8.0 * K * K * F + 30 * 8.0 * N * K * K
    print('lasso ista c128 %dx%d k=%d, 30 iterations: %.2f ms  %.1f real TFLOP/s' % (N, F, K, dt * 1e3, flops / dt / 1e12))


if __name__ == '__main__':
    main()
    complex128_lasso()
