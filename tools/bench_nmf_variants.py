#!/usr/bin/env python3
"""Secondary measurement: one MU iteration of the masked / KL variants (BASELINE configs[3] is
masked NMF at 131072 x 4096, k = 256 over 8 GPUs = 16384 rows per GPU).  Per-kernel-group
hipEvent timings through dcp_profile_*.   python tools/bench_nmf_variants.py [--rows 16384]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
from decomp_amd import _arrays, _hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=16384)
    ap.add_argument('--f', type=int, default=4096)
    ap.add_argument('--k', type=int, default=256)
    ap.add_argument('--steps', type=int, default=20)
    a = ap.parse_args()
    N, F, K = a.rows, a.f, a.k
    g = torch.Generator(device='cuda')
    g.manual_seed(1)
    Dt = torch.randn((K, F), generator=g, device='cuda').clamp_(min=0)
    xt = torch.randn((N, K), generator=g, device='cuda').clamp_(min=0)
    Y = xt @ Dt + 0.1 * torch.randn((N, F), generator=g, device='cuda').abs_()
    D0 = (Dt + 0.3 * torch.randn((K, F), generator=g, device='cuda')).clamp_(min=0.1)
    mask = (torch.rand((N, F), generator=g, device='cuda') >= 0.2).float()
    del xt
    lib, h = _arrays.lib_handle(Y)
    for name, lik, m in (('l2', 0, None), ('l2+mask', 0, mask), ('kl', 1, None), ('kl+mask', 1, mask)):
        D = D0.clone()
        _arrays.l2_normalize_(D, strict=True)
        x = torch.ones((N, K), device='cuda')
        it = ctypes.c_int(0)

        def run(n):
            _hip.check(h, lib.dcp_nmf_mu_f32(h, _arrays.ptr(Y), _arrays.ptr(m), _arrays.ptr(x),
                                             _arrays.ptr(D), N, F, K, lik, ctypes.c_float(0.0), n + 1,
                                             ctypes.byref(it), None, None), 'nmf')
        run(3)
        lib.dcp_profile_reset(h)
        lib.dcp_profile_enable(h, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(a.steps)
        e1.record()
        torch.cuda.synchronize()
        lib.dcp_profile_enable(h, 0)
        ms = e0.elapsed_time(e1) / a.steps
        prof = {}
        for lab in range(_hip.PROF_NLABELS):
            t, c = ctypes.c_double(0), ctypes.c_int64(0)
            lib.dcp_profile_read(h, lab, ctypes.byref(t), ctypes.byref(c))
            if c.value:
                prof[lib.dcp_profile_label_name(lab).decode()] = round(t.value / a.steps, 3)
        flops = (4.0 * N * K * F + 4.0 * N * K * K + 4.0 * K * K * F) if name == 'l2' else 12.0 * N * K * F
        if name.startswith('kl'):
            flops = 8.0 * N * K * F + (4.0 * N * K * F if m is not None else 0)
        print('%-8s rows=%d: %.3f ms/iter  %.1f TFLOP/s (algorithmic)  finite=%s  per-iteration ms: %s'
              % (name, N, ms, flops / ms / 1e9, bool(torch.isfinite(D).all()), prof))


if __name__ == '__main__':
    main()
