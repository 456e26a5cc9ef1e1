#!/usr/bin/env python3
"""Time the GEMM cores on the shapes of the NMF MU step (tuning aid; run on the GPU box).

    python tools/gemm_sweep.py [--n 65536 --f 4096 --k 256] [--reps 5]
"""
import argparse
import ctypes
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
from decomp_amd import _arrays, _hip  # noqa: E402


def time_gemm(form, M, N, K, ksplits, tile, reps):
    if form == 0:
        A = torch.rand((M, K), device='cuda') - 0.5
        B = torch.rand((N, K), device='cuda') - 0.5
    elif form == 1:
        A = torch.rand((M, K), device='cuda') - 0.5
        B = torch.rand((K, N), device='cuda') - 0.5
    else:
        A = torch.rand((K, M), device='cuda') - 0.5
        B = torch.rand((K, N), device='cuda') - 0.5
    C = torch.empty((M, N), device='cuda')
    lib, h = _arrays.lib_handle(A)
    fn = lib.dcp_gemm_f32

    def run():
        _hip.check(h, fn(h, form, _arrays.ptr(A), _arrays.ptr(B), _arrays.ptr(C), M, N, K,
                         ksplits, tile), 'gemm')
    run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=65536)
    ap.add_argument('--f', type=int, default=4096)
    ap.add_argument('--k', type=int, default=256)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--tiles', default='1,3,4,5,6,7,8,9')
    a = ap.parse_args()
    N, F, K = a.n, a.f, a.k
    rows = []
    tiles = [int(t) for t in a.tiles.split(',')]
    # P = Y D^T : NT [N,F]x[K,F]
    for tile in tiles:
        ms = time_gemm(0, N, K, F, 1, tile, a.reps)
        rows.append(('NT Y.D^T', tile, 1, ms, 2.0 * N * K * F / ms / 1e9))
    # stats = x^T [Y] : TN, split-K
    for tile in tiles:
        for ks in (16, 30):
            ms = time_gemm(2, K, F, N, ks, tile, a.reps)
            rows.append(('TN x^T.Y', tile, ks, ms, 2.0 * N * K * F / ms / 1e9))
    # Q = x G : NN [N,K]x[K,K]
    for tile in tiles:
        ms = time_gemm(1, N, K, K, 1, tile, a.reps)
        rows.append(('NN x.G', tile, 1, ms, 2.0 * N * K * K / ms / 1e9))
    # f = x D : NN [N,K]x[K,F] (masked path)
    for tile in tiles:
        ms = time_gemm(1, N, F, K, 1, tile, a.reps)
        rows.append(('NN x.D', tile, 1, ms, 2.0 * N * K * F / ms / 1e9))
    # square reference point
    for tile in tiles:
        ms = time_gemm(0, 4096, 4096, 4096, 1, tile, a.reps)
        rows.append(('NT 4096^3', tile, 1, ms, 2.0 * 4096 ** 3 / ms / 1e9))
    print('%-12s %4s %6s %10s %10s' % ('gemm', 'tile', 'splits', 'ms', 'TFLOP/s'))
    for r in rows:
        print('%-12s %4d %6d %10.3f %10.1f' % r)


if __name__ == '__main__':
    main()
