"""float64 dictionary step at the configs[2] minibatch shape: python tools/bench_dl_f64.py"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from decomp_amd import _arrays, _hip
lib = _hip.load()
MB, F, K = 8192, 4096, 512
g = torch.Generator(device='cuda'); g.manual_seed(2)
dt = torch.float64
Dt = torch.randn((K, F), generator=g, device='cuda', dtype=dt)
xt = 30.0 * torch.randn((MB, K), generator=g, device='cuda', dtype=dt) * (torch.rand((MB, K), generator=g, device='cuda') < 0.05)
Y = xt @ Dt + 0.1 * torch.randn((MB, F), generator=g, device='cuda', dtype=dt)
D = Dt + 0.2 * torch.randn((K, F), generator=g, device='cuda', dtype=dt)
_arrays.l2_normalize_(D, strict=True)
x = torch.ones((MB, K), device='cuda', dtype=dt)
A = torch.zeros((K, K), device='cuda', dtype=dt); B = torch.zeros((K, F), device='cuda', dtype=dt)
Dn = torch.empty_like(D)
_, h = _arrays.lib_handle(D)
md, lit = ctypes.c_double(0), ctypes.c_int(0)
cnt = [0]
def step(method):
    global D, Dn
    theta = cnt[0] * MB + 1.0
    _hip.check(h, lib.dcp_dict_step_f64(h, _arrays.ptr(Y), _arrays.ptr(x), _arrays.ptr(D), _arrays.ptr(Dn), _arrays.ptr(A),
                                        _arrays.ptr(B), MB, F, K, (theta - MB) / theta, 0.1, method, 10, 1e-5,
                                        ctypes.byref(md), ctypes.byref(lit)), 'dict_step f64')
    D, Dn = Dn, D
    cnt[0] += 1
for name, m in (('ista', _hip.LASSO_ISTA), ('cd', _hip.LASSO_CD)):
    for _ in range(3): step(m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6): step(m)
    e1.record(); torch.cuda.synchronize()
    print('float64 dictionary step 8192x4096 k=512 %s x10: %.3f ms' % (name, e0.elapsed_time(e1) / 6))
