import torch
torch.backends.cuda.matmul.allow_tf32 = False
g = torch.Generator(device='cuda'); g.manual_seed(0)
def rnd(shape, dt):
    if dt.is_complex:
        return torch.complex(torch.rand(shape, generator=g, device='cuda'), torch.rand(shape, generator=g, device='cuda')).to(dt)
    return torch.rand(shape, generator=g, device='cuda').to(dt)
def t(label, A, B, trans_b, n=10):
    Bt = B.t() if trans_b else B
    fn = lambda: torch.mm(A, Bt)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    M, K = A.shape; N = Bt.shape[1]
    fl = (8.0 if A.dtype.is_complex else 2.0) * M * N * K
    print('%-56s %.3f ms  %.1f TF' % (label, ms, fl / ms / 1e9))
f32, f64, c64 = torch.float32, torch.float64, torch.complex64
t('f32 shard Y.D^T 8192x256x4096 (here 0.142 ms incl. split)', rnd((8192, 4096), f32), rnd((256, 4096), f32), True)
t('f32 shard x^T.Y 256x4096x8192 (here 0.155 ms)', rnd((8192, 256), f32).t(), rnd((8192, 4096), f32), False)
t('f32 ISTA x.G 8192x512x512 (here 0.050 ms with prox)', rnd((8192, 512), f32), rnd((512, 512), f32), False)
t('f32 y.A^T 8192x512x4096 (here 0.30 ms)', rnd((8192, 4096), f32), rnd((512, 4096), f32), True)
t('f32 x^T.y 512x4096x8192 (here 0.33 ms with x^T x)', rnd((8192, 512), f32).t(), rnd((8192, 4096), f32), False)
t('f32 k=32 Y.D^T 65536x32x4096 (here 0.22 ms)', rnd((65536, 4096), f32), rnd((32, 4096), f32), True)
t('f32 k=32 x^T.Y 32x4096x65536 (here 0.32 ms)', rnd((65536, 32), f32).t(), rnd((65536, 4096), f32), False)
t('f64 Y.D^T 32768x256x4096 (here 64 TF)', rnd((32768, 4096), f64), rnd((256, 4096), f64), True)
t('f64 x^T.Y 256x4096x32768', rnd((32768, 256), f64).t(), rnd((32768, 4096), f64), False)
t('c64 y.A^H 8192x512x8192 (here 2.17 ms)', rnd((8192, 8192), c64), rnd((512, 8192), c64).conj(), True)
t('c64 x^H.y 512x8192x8192 (here 2.29 ms with x^H x)', rnd((8192, 512), c64).conj().t(), rnd((8192, 8192), c64), False)
t('c64 x.G 8192x512x512 (here 0.167 ms with prox)', rnd((8192, 512), c64), rnd((512, 512), c64), False)
