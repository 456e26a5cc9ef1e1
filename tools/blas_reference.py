"""How fast does the vendor BLAS (through torch.mm, fp32, no TF32) run the two big products of the headline
step?  A yardstick for the hand-written kernels, nothing the library calls.  python tools/blas_reference.py"""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
N, F, K = 65536, 4096, 256
g = torch.Generator(device='cuda'); g.manual_seed(0)
Y = torch.rand((N, F), generator=g, device='cuda')
D = torch.rand((K, F), generator=g, device='cuda')
x = torch.rand((N, K), generator=g, device='cuda')
def t(fn, flops, label, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print('%-38s %.3f ms  %.1f TF' % (label, ms, flops / ms / 1e9))
out1 = torch.empty((N, K), device='cuda')
out2 = torch.empty((K, F), device='cuda')
t(lambda: torch.mm(Y, D.t(), out=out1), 2.0 * N * K * F, 'torch.mm  Y . D^T  (65536x256x4096)')
t(lambda: torch.mm(x.t(), Y, out=out2), 2.0 * N * K * F, 'torch.mm  x^T . Y  (256x4096x65536)')
