#!/usr/bin/env python3
"""Secondary measurement (BASELINE configs[2]): one epoch of dictionary_learning.solve at
Y 65536 x 4096, k = 512, alpha = 0.1, fp32, minibatch 8192, lasso ista x 10.
Prints per-step wall time and the per-phase split measured with hipEvents around the C-ABI
calls.  Run on the GPU box:  python tools/bench_dl.py [--lasso cd] [--k 512]"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402
from decomp_amd import _arrays, _hip, lasso as hl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=65536)
    ap.add_argument('--f', type=int, default=4096)
    ap.add_argument('--k', type=int, default=512)
    ap.add_argument('--mb', type=int, default=8192)
    ap.add_argument('--lasso', default='ista')
    ap.add_argument('--lasso-iter', type=int, default=10)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--complex', action='store_true', help='complex64 (BASELINE configs[4] shape: --f 8192)')
    a = ap.parse_args()
    N, F, K, MB = a.n, a.f, a.k, a.mb
    g = torch.Generator(device='cuda')
    g.manual_seed(2)
    dt = torch.complex64 if a.complex else torch.float32

    def randn(*s):
        r = torch.randn(s, generator=g, device='cuda')
        if a.complex:
            return torch.complex(r, torch.randn(s, generator=g, device='cuda'))
        return r
    Dt = randn(K, F)
    xt = 30.0 * randn(N, K) * (torch.rand((N, K), generator=g, device='cuda') < 0.05)
    Y = xt @ Dt + 0.1 * randn(N, F)
    D = Dt + 0.2 * randn(K, F)
    del xt
    x = torch.ones((N, K), device='cuda', dtype=dt)
    _arrays.l2_normalize_(D, strict=True)
    A = torch.zeros((K, K), device='cuda', dtype=dt)
    B = torch.zeros((K, F), device='cuda', dtype=dt)
    D_new = torch.empty_like(D)
    stats = torch.empty((K, F + K), device='cuda', dtype=dt)
    md = torch.zeros((1,), device='cuda')
    sfx = 'c64' if a.complex else 'f32'
    fn_stats = getattr(lib_ := _hip.load(), 'dcp_dict_stats_' + sfx)
    fn_update = getattr(lib_, 'dcp_dict_update_' + sfx)
    lib, h = _arrays.lib_handle(D)
    code = hl._METHOD_CODE[a.lasso]
    lasso_it = ctypes.c_int(0)

    def step(r, count, timed):
        nonlocal D, D_new
        y_mb = Y[r * MB:(r + 1) * MB]
        x_mb = x[r * MB:(r + 1) * MB]
        theta = count * MB + 1.0
        beta = (theta - MB) / theta
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        _hip.check(h, fn_stats(h, _arrays.ptr(y_mb), _arrays.ptr(x_mb), _arrays.ptr(D),
                                             MB, F, K, 0.1, code, a.lasso_iter, 1e-5,
                                             _arrays.ptr(stats), ctypes.byref(lasso_it)), 'stats')
        ev[1].record()
        _hip.check(h, fn_update(h, _arrays.ptr(stats), beta, _arrays.ptr(A),
                                              _arrays.ptr(B), _arrays.ptr(D), _arrays.ptr(D_new),
                                              F, K, _arrays.ptr(md)), 'update')
        ev[2].record()
        torch.cuda.synchronize()
        D, D_new = D_new, D
        return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])

    n_mb = N // MB
    step(0, 0, False)
    t0 = time.perf_counter()
    acc = np.zeros(2)
    for s in range(a.steps):
        acc += np.array(step((s + 1) % n_mb, s + 1, True))
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    nnz = float((x[:MB] != 0).float().mean())
    flops = 2.0 * MB * F * K + 2.0 * K * K * F + a.lasso_iter * 2.0 * MB * K * K + \
        2.0 * MB * K * K + 2.0 * MB * K * F + 2.0 * K * K * F
    if a.complex:
        flops *= 4.0
    print('dictionary step %s N_mb=%d F=%d K=%d lasso=%s x%d : %.3f ms/step wall '
          '(lasso+stats %.3f ms, A/B + atom sweep + max|dD| %.3f ms)  %.1f TFLOP/s algorithmic, '
          'code density %.3f, finite D: %s'
          % (sfx, MB, F, K, a.lasso, a.lasso_iter, wall, acc[0] / a.steps, acc[1] / a.steps,
             flops / wall / 1e9, nnz, bool(torch.isfinite(D).all())))


if __name__ == '__main__':
    main()
