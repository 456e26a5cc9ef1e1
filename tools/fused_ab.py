"""A/B of the fused D-side launch against the separate launches, interleaved in ONE process (same box, same
clocks): ms per MU iteration of dcp_nmf_mu_f32 for several shapes.  DCP_NO_FUSED_UPDATE is read per call."""
import ctypes, os, sys, time
sys.path.insert(0, '.')
import torch
from decomp_amd import _arrays, _hip

def run(lib, h, Y, x, D, n):
    it = ctypes.c_int(0)
    N, F = Y.shape; K = D.shape[0]
    _hip.check(h, lib.dcp_nmf_mu_f32(h, _arrays.ptr(Y), None, _arrays.ptr(x), _arrays.ptr(D), N, F, K, 0,
                                     ctypes.c_float(0.0), n + 1, ctypes.byref(it), None, None), 'mu')

for (N, F, K) in [(8192, 4096, 256), (65536, 4096, 256), (2048, 512, 32), (8192, 1024, 64), (256, 128, 8), (16384, 4096, 128)]:
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    Y = torch.rand((N, F), generator=g, device='cuda')
    D = torch.rand((K, F), generator=g, device='cuda') + 0.1
    _arrays.l2_normalize_(D, strict=True)
    lib, h = _arrays.lib_handle(D)
    res = {}
    n = 200 if N * F < 1e8 else 50
    for rnd in range(4):
        for mode in ('fused', 'fused_mdcopy', 'separate'):
            os.environ.pop('DCP_NO_FUSED_UPDATE', None); os.environ.pop('DCP_FUSED_MD_DEVICE', None)
            if mode == 'separate': os.environ['DCP_NO_FUSED_UPDATE'] = '1'
            if mode == 'fused_mdcopy': os.environ['DCP_FUSED_MD_DEVICE'] = '1'
            x = torch.ones((N, K), device='cuda'); Dc = D.clone()
            run(lib, h, Y, x, Dc, 5); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(lib, h, Y, x, Dc, n); torch.cuda.synchronize()
            res.setdefault(mode, []).append((time.perf_counter() - t0) / n * 1e6)
    print('%6dx%5d k=%3d  ' % (N, F, K) + '  '.join('%s %s us' % (m, ' '.join('%.1f' % v for v in vs)) for m, vs in res.items()), flush=True)
