"""Analysis: does running the x update + statistics of an MU iteration row chunk by row chunk (so that the second read of
a chunk of Y hits the Infinity Cache) beat one pass over all rows?  dcp_nmf_mu_stats_f32 on [rows, 4096] with k atoms,
whole vs in 2 / 4 / 8 chunks (each call = Gram + x.G + x update + x^T [Y | x] of its rows).
python tools/mall_chunk_probe.py [rows] [k]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decomp_amd import _arrays, _hip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
F = 4096
g = torch.Generator(device='cuda'); g.manual_seed(0)
Y = torch.rand((N, F), generator=g, device='cuda')
D = torch.rand((K, F), generator=g, device='cuda') + 0.1
_arrays.l2_normalize_(D, strict=True)
x = torch.ones((N, K), device='cuda')
xo = torch.empty_like(x)
lib, h = _arrays.lib_handle(D)
W = lib.dcp_nmf_mu_stats_width(F, K, 0, 0)
stats = torch.empty((8, K, W), device='cuda')


def run(chunks):
    rows = N // chunks
    for c in range(chunks):
        r0 = c * rows
        _hip.check(h, lib.dcp_nmf_mu_stats_f32(h, _arrays.ptr(Y[r0:r0 + rows]), None, _arrays.ptr(x[r0:r0 + rows]),
                                               _arrays.ptr(xo[r0:r0 + rows]), _arrays.ptr(D), rows, F, K, 0,
                                               _arrays.ptr(stats[c])), 'stats')


for chunks in (1, 2, 4, 8):
    for _ in range(5):
        run(chunks)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(chunks)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print('rows %d k %d chunks %d: %.1f us per pass (Y = %.0f MB; two reads at %.2f TB/s)' % (
        N, K, chunks, 1e3 * best, N * F * 4 / 1e6, 2.0 * N * F * 4 / best / 1e9))
