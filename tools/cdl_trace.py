"""configs[4] complex64 dictionary step for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 tools/cdl_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from decomp_amd import _arrays, _hip
lib = _hip.load()
device = 'cuda'
MB, F, K = 8192, 8192, 512
g = torch.Generator(device=device); g.manual_seed(3)
def crandn(*sh):
    return torch.complex(torch.randn(sh, generator=g, device=device), torch.randn(sh, generator=g, device=device))
Dt = crandn(K, F)
xt = 30.0 * crandn(MB, K) * (torch.rand((MB, K), generator=g, device=device) < 0.05)
Yc = xt @ Dt + 0.1 * crandn(MB, F)
Dc = Dt + 0.2 * crandn(K, F)
del xt, Dt
_arrays.l2_normalize_(Dc, strict=True)
xc = torch.ones((MB, K), device=device, dtype=torch.complex64)
Ac = torch.zeros((K, K), device=device, dtype=torch.complex64)
Bc = torch.zeros((K, F), device=device, dtype=torch.complex64)
Dn = torch.empty_like(Dc)
_, h = _arrays.lib_handle(Yc)
md = ctypes.c_double(0); lit = ctypes.c_int(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n_steps = int(os.environ.get('STEPS', 6))
for i in range(n_steps):
    if i == 2:
        e0.record()
    theta = i * MB + 1.0
    _hip.check(h, lib.dcp_dict_step_c64(h, _arrays.ptr(Yc), _arrays.ptr(xc), _arrays.ptr(Dc), _arrays.ptr(Dn),
                                        _arrays.ptr(Ac), _arrays.ptr(Bc), MB, F, K, (theta - MB) / theta, 0.1,
                                        _hip.LASSO_ISTA, 10, 1e-5, ctypes.byref(md), ctypes.byref(lit)), 'dict_step c64')
    Dc, Dn = Dn, Dc
e1.record()
torch.cuda.synchronize()
print('complex64 dictionary step %.4f ms' % (e0.elapsed_time(e1) / (n_steps - 2)))
