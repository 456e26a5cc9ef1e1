"""Analysis: the bare dictionary step (dcp_dict_step_async_f32, configs[2]) on the SAME minibatch every step (what
bench.py's dictionary_step_ms times) against a DIFFERENT contiguous 8192-row slice of a 65536-row Y every step (what a
step inside dictionary_learning.solve sees, minus its gathers): is the end-to-end gap data freshness (Infinity Cache
/ L2 residency of the minibatch) or overhead?   python tools/dl_fresh.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from decomp_amd import _arrays, _hip
lib = _hip.load()
dev = 'cuda'
g = torch.Generator(device=dev); g.manual_seed(2)
NT, MB, F, K = 65536, 8192, 4096, 512
Dt = torch.randn((K, F), generator=g, device=dev)
Y = torch.empty((NT, F), device=dev)
for r0 in range(0, NT, MB):
    xt = 30.0 * torch.randn((MB, K), generator=g, device=dev) * (torch.rand((MB, K), generator=g, device=dev) < 0.05)
    Y[r0:r0 + MB] = xt @ Dt + 0.1 * torch.randn((MB, F), generator=g, device=dev)
D = Dt + 0.2 * torch.randn((K, F), generator=g, device=dev)
_arrays.l2_normalize_(D, strict=True)
xall = torch.ones((NT, K), device=dev)
A = torch.zeros((K, K), device=dev); B = torch.zeros((K, F), device=dev); Dn = torch.empty_like(D)
_, h = _arrays.lib_handle(Y)
md_pin = torch.zeros((2,), dtype=torch.float32).pin_memory(); md_np = md_pin.numpy()
lit = ctypes.c_int(0)
state = {'c': 0, 'D': D, 'Dn': Dn}

def step(fresh):
    c = state['c']
    r0 = (c % (NT // MB)) * MB if fresh else 0
    theta = c * MB + 1.0
    md_np[c & 1] = -1.0
    _hip.check(h, lib.dcp_dict_step_async_f32(h, _arrays.ptr(Y[r0:r0 + MB]), _arrays.ptr(xall[r0:r0 + MB]), _arrays.ptr(state['D']),
                                              _arrays.ptr(state['Dn']), _arrays.ptr(A), _arrays.ptr(B), MB, F, K,
                                              (theta - MB) / theta, 0.1, _hip.LASSO_ISTA, 10, 1e-5,
                                              _arrays.ptr(md_pin[(c & 1):(c & 1) + 1]), ctypes.byref(lit)), 'step')
    if c > 0:
        while md_np[(c & 1) ^ 1] == -1.0:
            pass
    state['D'], state['Dn'] = state['Dn'], state['D']
    state['c'] = c + 1

for fresh in (False, True, False, True):
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        step(fresh)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(48):
        step(fresh)
    e1.record(); torch.cuda.synchronize()
    print('%s minibatch every step: %.4f ms per step' % ('a DIFFERENT' if fresh else 'the SAME', e0.elapsed_time(e1) / 48), flush=True)
