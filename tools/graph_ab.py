"""Does replaying a captured hipGraph of two MU iterations beat launching their kernels one by one?
Small / medium problems (the loop is launch bound there).  python tools/graph_ab.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import torch
from decomp_amd import _arrays, _hip

lib = _hip.load()
for (N, F, K) in [(256, 128, 8), (2048, 512, 32), (8192, 1024, 64), (8192, 4096, 256)]:
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    Y = torch.rand((N, F), generator=g, device='cuda')
    D = torch.rand((K, F), generator=g, device='cuda') + 0.1
    _arrays.l2_normalize_(D, strict=True)
    D2 = torch.empty_like(D)
    x = torch.ones((N, K), device='cuda'); x2 = torch.empty_like(x)
    W = lib.dcp_nmf_mu_stats_width(F, K, 0, 0)
    stats = torch.empty((K, W), device='cuda')
    md = torch.zeros((2,), device='cuda')
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        _, h = _arrays.lib_handle(D)

        def half(xa, xb, Da, Db, slot):
            _hip.check(h, lib.dcp_nmf_mu_stats_f32(h, _arrays.ptr(Y), None, _arrays.ptr(xa), _arrays.ptr(xb),
                                                   _arrays.ptr(Da), N, F, K, 0, _arrays.ptr(stats)), 'stats')
            _hip.check(h, lib.dcp_nmf_mu_update_f32(h, _arrays.ptr(stats), _arrays.ptr(Da), _arrays.ptr(Db), F, K, 0, 0,
                                                    _arrays.ptr(md[slot:slot + 1]),
                                                    _arrays.ptr(md[slot ^ 1:(slot ^ 1) + 1])), 'update')

        def pair():
            half(x, x2, D, D2, 0)
            half(x2, x, D2, D, 1)
        for _ in range(3):
            pair()
        s.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            pair()
        s.synchronize()
        plain = (time.perf_counter() - t0) / (2 * n) * 1e6
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            pair()
        for _ in range(3):
            graph.replay()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            graph.replay()
        s.synchronize()
        rep = (time.perf_counter() - t0) / (2 * n) * 1e6
        it = ctypes.c_int(0)
        xs = torch.ones((N, K), device='cuda')
        Ds = D.clone()

        def solve(n_it):
            _hip.check(h, lib.dcp_nmf_mu_f32(h, _arrays.ptr(Y), None, _arrays.ptr(xs), _arrays.ptr(Ds), N, F, K, 0,
                                             ctypes.c_float(0.0), n_it, ctypes.byref(it), None, None), 'nmf')
        solve(5)
        s.synchronize()
        t0 = time.perf_counter()
        solve(2 * n + 1)
        s.synchronize()
        cloop = (time.perf_counter() - t0) / (2 * n) * 1e6
    print('%5d x %4d k=%3d: launches %.1f us/iteration, graph replay %.1f us/iteration, C loop with lagged stop test %.1f us/iteration'
          % (N, F, K, plain, rep, cloop))
