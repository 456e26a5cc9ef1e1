"""One dictionary-learning minibatch step for a kernel trace (BASELINE configs[2] shape by default):
rocprofv3 --kernel-trace --stats -- python3 tools/dl_trace.py   (env: STEPS, METHOD = ista | cd, CPLX = 1, F, K, MB)"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from decomp_amd import _arrays, _hip
lib = _hip.load()
device = 'cuda'
cplx = os.environ.get('CPLX', '0') == '1'
MB, F, K = int(os.environ.get('MB', 8192)), int(os.environ.get('F', 8192 if cplx else 4096)), int(os.environ.get('K', 512))
method = {'ista': _hip.LASSO_ISTA, 'cd': _hip.LASSO_CD, 'fista': _hip.LASSO_FISTA}[os.environ.get('METHOD', 'ista')]
g = torch.Generator(device=device); g.manual_seed(2)
def randn(*sh):
    r = torch.randn(sh, generator=g, device=device)
    return torch.complex(r, torch.randn(sh, generator=g, device=device)) if cplx else r
Dt = randn(K, F)
xt = 30.0 * randn(MB, K) * (torch.rand((MB, K), generator=g, device=device) < 0.05)
Y = xt @ Dt + 0.1 * randn(MB, F)
D = Dt + 0.2 * randn(K, F)
del xt, Dt
_arrays.l2_normalize_(D, strict=True)
dt = torch.complex64 if cplx else torch.float32
x = torch.ones((MB, K), device=device, dtype=dt)
A = torch.zeros((K, K), device=device, dtype=dt)
B = torch.zeros((K, F), device=device, dtype=dt)
Dn = torch.empty_like(D)
_, h = _arrays.lib_handle(Y)
sfx = 'c64' if cplx else 'f32'
use_async = os.environ.get('ASYNC', '1') == '1'      # the entry dictionary_learning.solve() drives (max|dD| read a step late)
fn = getattr(lib, ('dcp_dict_step_async_' if use_async else 'dcp_dict_step_') + sfx)
md = ctypes.c_double(0); lit = ctypes.c_int(0)
md_pin = torch.zeros((2,), dtype=torch.float32).pin_memory()
md_np = md_pin.numpy()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n_steps = int(os.environ.get('STEPS', 8))
for i in range(n_steps):
    if i == 2:
        e0.record()
    theta = i * MB + 1.0
    if use_async:
        md_np[i & 1] = -1.0
        out = _arrays.ptr(md_pin[(i & 1):(i & 1) + 1])
    else:
        out = ctypes.byref(md)
    _hip.check(h, fn(h, _arrays.ptr(Y), _arrays.ptr(x), _arrays.ptr(D), _arrays.ptr(Dn), _arrays.ptr(A), _arrays.ptr(B),
                     MB, F, K, (theta - MB) / theta, 0.1, method, 10, 1e-5, out, ctypes.byref(lit)), 'dict_step')
    if use_async and i > 0:
        while md_np[(i & 1) ^ 1] == -1.0:
            pass
    D, Dn = Dn, D
e1.record()
torch.cuda.synchronize()
print('dictionary step (%s, %s, %s entry) %dx%d k=%d: %.4f ms' % (sfx, os.environ.get('METHOD', 'ista'),
                                                                   'async' if use_async else 'blocking', MB, F, K,
                                                                   e0.elapsed_time(e1) / (n_steps - 2)))
