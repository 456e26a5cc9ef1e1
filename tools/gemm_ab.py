#!/usr/bin/env python3
"""Interleaved A/B of GEMM tile variants in ONE process on ONE device (guide rule 24), on
data shaped like the NMF step's (non-negative, low-rank + noise) or uniform random.
    python tools/gemm_ab.py --form 0 --tiles 7,10,11 --rounds 8 [--positive]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
from decomp_amd import _arrays, _hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--form', type=int, default=0)
    ap.add_argument('--m', type=int, default=65536)
    ap.add_argument('--n', type=int, default=256)
    ap.add_argument('--k', type=int, default=4096)
    ap.add_argument('--splits', type=int, default=1)
    ap.add_argument('--tiles', default='1,7,10,11')
    ap.add_argument('--rounds', type=int, default=8)
    ap.add_argument('--positive', action='store_true')
    a = ap.parse_args()
    M, N, K = a.m, a.n, a.k
    sh = {0: ((M, K), (N, K)), 1: ((M, K), (K, N)), 2: ((K, M), (K, N))}[a.form]

    def mk(shape):
        t = torch.rand(shape, device='cuda')
        if a.positive:
            return t * 3.0
        return t - 0.5
    A, B = mk(sh[0]), mk(sh[1])
    C = torch.empty((M, N), device='cuda')
    lib, h = _arrays.lib_handle(A)
    tiles = [int(t) for t in a.tiles.split(',')]
    times = {t: [] for t in tiles}

    def run(tile):
        _hip.check(h, lib.dcp_gemm_f32(h, a.form, _arrays.ptr(A), _arrays.ptr(B), _arrays.ptr(C), M, N, K,
                                       a.splits, tile), 'gemm')
    for t in tiles:
        run(t)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for t in tiles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(t); run(t); run(t)
            e1.record()
            torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) / 3)
    flops = 2.0 * M * N * K
    for t in tiles:
        v = sorted(times[t])
        med = v[len(v) // 2]
        print('form %d tile %2d: median %.3f ms (%.1f TF)  min %.3f ms (%.1f TF)'
              % (a.form, t, med, flops / med / 1e9, v[0], flops / v[0] / 1e9))


if __name__ == '__main__':
    main()
