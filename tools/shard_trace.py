"""One 8192-row shard of configs[1] for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 tools/shard_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from decomp_amd import _arrays, _hip
lib = _hip.load()
N, F, K = int(os.environ.get('ROWS', 8192)), 4096, 256
g = torch.Generator(device='cuda'); g.manual_seed(1)
Y = torch.rand((N, F), generator=g, device='cuda')
D = torch.rand((K, F), generator=g, device='cuda') + 0.1
_arrays.l2_normalize_(D, strict=True)
x = torch.ones((N, K), device='cuda')
_, h = _arrays.lib_handle(Y)
it = ctypes.c_int(0)
for n_it in (3, 101):
    _hip.check(h, lib.dcp_nmf_mu_f32(h, _arrays.ptr(Y), None, _arrays.ptr(x), _arrays.ptr(D), N, F, K, 0,
                                     ctypes.c_float(0.0), n_it, ctypes.byref(it), None, None), 'nmf')
torch.cuda.synchronize()
