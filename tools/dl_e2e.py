"""Analysis: decomp_amd.dictionary_learning.solve end to end at configs[2] (65536 x 4096, k = 512, minibatch 8192,
ista x 10): wall clock per step, run-to-run determinism.  Usage: python tools/dl_e2e.py [epochs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import decomp_amd
g = torch.Generator(device='cuda'); g.manual_seed(2)
rows, F, K, MB = 65536, 4096, 512, 8192
Dt = torch.randn((K, F), generator=g, device='cuda')
xt = 30.0 * torch.randn((rows, K), generator=g, device='cuda') * (torch.rand((rows, K), generator=g, device='cuda') < 0.05)
Y = xt @ Dt + 0.1 * torch.randn((rows, F), generator=g, device='cuda')
D0 = Dt + 0.2 * torch.randn((K, F), generator=g, device='cuda')
del xt
ep = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kw = dict(tol=0.0, minibatch=MB, maxiter=ep + 1, lasso_method=os.environ.get('LM', 'ista'), lasso_iter=10, lasso_tol=1e-5, random_seed=0)
first = decomp_amd.dictionary_learning.solve(Y, D0, 0.1, **kw)      # cold: straight after the data synthesis
res = []
for rep in range(int(os.environ.get('REPS', '4'))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if os.environ.get('DCP_DL_TRACE', '0') == '1':
        print('call starts at 0', flush=True)
        import decomp_amd.dictionary_learning as _dl
        _orig = _dl.solve_cd_indexed
        def _stamped(*a, **k):
            print('solve_cd_indexed entered after %.3f ms' % (1e3 * (time.perf_counter() - t0)), flush=True)
            r = _orig(*a, **k)
            print('solve_cd_indexed returned after %.3f ms' % (1e3 * (time.perf_counter() - t0)), flush=True)
            return r
        _dl.solve_cd_indexed = _stamped
    it, D, x = decomp_amd.dictionary_learning.solve(Y, D0, 0.1, **kw)
    t_ret = time.perf_counter()
    if os.environ.get('DCP_DL_TRACE', '0') == '1':
        _dl.solve_cd_indexed = _orig
    torch.cuda.synchronize()
    if os.environ.get('DCP_DL_TRACE', '0') == '1':
        print('solve() returned after %.3f ms, final sync %.3f ms' % (1e3 * (t_ret - t0), 1e3 * (time.perf_counter() - t_ret)), flush=True)
    ms = 1e3 * (time.perf_counter() - t0) / (ep * (rows // MB))
    print('rep', rep, 'prefetch', os.environ.get('DCP_DL_PREFETCH', '1'), 'ms/step %.4f' % ms,
          'same as cold first call: D', bool(torch.equal(D, first[1])), 'x', bool(torch.equal(x, first[2])), flush=True)
