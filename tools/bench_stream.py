#!/usr/bin/env python3
"""Out-of-core feed: AsyncMinibatchData streaming host data through the GPU.

(a) the reference's own container benchmark shape (speed_tests/tests/test_data.py:6-22 scaled to
    fit: rows x 100000 fp32, minibatch 10, n_parallel 3, `arr *= 2`), read and read+write;
(b) one epoch of dictionary learning at the BASELINE configs[2] shape (Y 65536 x 4096, k = 512,
    minibatch 8192, ista x 10) with Y and x in pinned host memory vs everything in HBM.
Run on the GPU box:  python tools/bench_stream.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import decomp_amd  # noqa: E402
from decomp_amd.utils.data import AsyncMinibatchData  # noqa: E402


def container_rate(rows, cols, mb, needs_update, n_parallel=3, big=False):
    a = np.ones((rows, cols), np.float32)
    data = AsyncMinibatchData(a, mb, n_parallel=n_parallel, needs_update=needs_update)
    for _ in range(2):
        t0 = time.perf_counter()
        for arr in data:
            arr.mul_(2.0)
        data.flush()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    nbytes = data.n_loop * mb * cols * 4 * (2 if needs_update else 1)
    print('container %6d x %6d mb %5d  %-10s : %7.1f ms/epoch  %6.1f GB/s over PCIe  (%d rounds, %.1f us/round)'
          % (rows, cols, mb, 'read+write' if needs_update else 'read', dt * 1e3, nbytes / dt / 1e9,
             data.n_loop, dt / data.n_loop * 1e6))


def dl_epoch(streamed):
    N, F, K, MB = 65536, 4096, 512, 8192
    g = torch.Generator(device='cuda')
    g.manual_seed(2)
    Dt = torch.randn((K, F), generator=g, device='cuda')
    xt = 30.0 * torch.randn((N, K), generator=g, device='cuda') * (torch.rand((N, K), generator=g, device='cuda') < 0.05)
    Y = xt @ Dt + 0.1 * torch.randn((N, F), generator=g, device='cuda')
    D0 = Dt + 0.2 * torch.randn((K, F), generator=g, device='cuda')
    x0 = torch.ones((N, K), device='cuda')
    del xt
    if streamed:
        Y = Y.cpu().numpy()
        x0 = x0.cpu().numpy()
    torch.cuda.synchronize()
    best = {}
    for maxiter in (3, 7, 3, 7):                   # 2 and 6 epochs: the difference is steady state
        t0 = time.perf_counter()
        it, D, x = decomp_amd.dictionary_learning.solve(Y, D0, 0.1, x0, tol=0.0, minibatch=MB,
                                                        maxiter=maxiter, lasso_method='ista',
                                                        lasso_iter=10, lasso_tol=1e-5, random_seed=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best[maxiter] = min(best.get(maxiter, dt), dt)
    steps = 4 * (N // MB)
    per_step = (best[7] - best[3]) / steps
    print('dictionary learning, %s: 2 epochs %.1f ms, 6 epochs %.1f ms -> %.2f ms/step steady state '
          '(incl. per-epoch shuffle), set-up %.1f ms'
          % ('Y, x streamed from pinned host memory' if streamed else 'everything in HBM',
             best[3] * 1e3, best[7] * 1e3, per_step * 1e3, (best[3] - per_step * steps / 2) * 1e3))


if __name__ == '__main__':
    if '--dl-streamed-only' in sys.argv:
        dl_epoch(True)
        sys.exit(0)
    container_rate(20000, 100000, 10, False)
    container_rate(20000, 100000, 10, True)
    container_rate(65536, 4096, 8192, False)
    container_rate(65536, 4096, 8192, True)
    dl_epoch(False)
    dl_epoch(True)
