"""Analysis: the bare x.G product of one ISTA iteration (NN, M x K x K float32) on every tile code of the test hook.
python tools/ista_tile_sweep.py [M] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decomp_amd import _arrays, _hip
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
A = torch.rand((M, K), device='cuda') - 0.5
B = torch.rand((K, K), device='cuda') - 0.5
C = torch.empty((M, K), device='cuda')
lib, h = _arrays.lib_handle(A)
res = []
for tile in range(0, 35):
    def run():
        return lib.dcp_gemm_f32(h, 1, _arrays.ptr(A), _arrays.ptr(B), _arrays.ptr(C), M, K, K, 1, tile)
    if run() != 0:
        continue
    torch.cuda.synchronize()
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    res.append((best, tile))
for b, t in sorted(res):
    print('tile %2d  %.1f us  %.1f TF' % (t, 1e3 * b, 2.0 * M * K * K / b / 1e9))
