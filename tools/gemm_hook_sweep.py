"""Analysis: one product on every tile code x split count of the test hook (time includes the ordered slab sum).
python tools/gemm_hook_sweep.py <form 0=NT 1=NN 2=TN> <M> <N> <K> [splits e.g. 1,2,4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decomp_amd import _arrays, _hip
form, M, N, K = (int(v) for v in sys.argv[1:5])
splits = [int(v) for v in sys.argv[5].split(',')] if len(sys.argv) > 5 else [1]
shapeA = (M, K) if form != 2 else (K, M)
shapeB = (N, K) if form == 0 else (K, N)
A = torch.rand(shapeA, device='cuda') - 0.5
B = torch.rand(shapeB, device='cuda') - 0.5
C = torch.empty((M, N), device='cuda')
lib, h = _arrays.lib_handle(A)
res = []
for ks in splits:
    for tile in range(0, 35):
        def run():
            return lib.dcp_gemm_f32(h, form, _arrays.ptr(A), _arrays.ptr(B), _arrays.ptr(C), M, N, K, ks, tile)
        if run() != 0:
            continue
        torch.cuda.synchronize()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                run()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 8)
        res.append((best, tile, ks))
for b, t, ks in sorted(res)[:12]:
    print('tile %2d splits %2d  %.1f us  %.1f TF' % (t, ks, 1e3 * b, 2.0 * M * N * K / b / 1e9))
