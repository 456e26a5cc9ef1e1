"""GPU: every tile tier of the GEMM dispatch against the NumPy oracle (oracle.nmf.mu_step, the
reference's formulation: grads.py:108-125 + batch_mu.py:16-24) -- ranks from 8 to 260 atoms select
the 128x32 / 32x128, 64x64, 128x128 and 256x256 tiles (float32) and the narrow fp64 MFMA tiles
(float64); row / channel counts off the tile grid select the bounds-checked instantiation with its
clamped 16-byte loads, odd leading dimensions its element-wise loads.  Three MU iterations, the
per-iteration residual and D compared with the oracle's."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (rows, channels, atoms)
SHAPES = [(4096, 1024, 8), (4096, 1024, 20), (4096, 1024, 32), (4096, 1024, 50), (4096, 1024, 64),
          (4096, 1024, 100), (4096, 1024, 128), (4096, 1024, 250), (4096, 1024, 260),
          (4000, 1000, 32), (4100, 1028, 64), (4099, 1021, 30), (8192, 516, 256)]


def _problem(N, F, K, dtype, masked, seed):
    rng = np.random.RandomState(seed)
    xt = np.maximum(rng.randn(N, K), 0)
    Dt = np.maximum(rng.randn(K, F), 0)
    y = (xt.dot(Dt) + 0.1 * np.abs(rng.randn(N, F))).astype(dtype)
    d0 = np.maximum(Dt + 0.3 * rng.randn(K, F), 0.1).astype(dtype)
    m = (rng.rand(N, F) >= 0.2).astype(dtype) if masked else None
    return y, d0, m


def _residual(y, x, d, m):
    r = y.astype(np.float64) - x.astype(np.float64).dot(d.astype(np.float64))
    if m is not None:
        r = r * m
    return float(np.sqrt(np.sum(r * r)))


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('shape', [(4096, 1024, 8), (4096, 1024, 50), (4099, 1021, 30), (4096, 1024, 128), (4100, 1028, 64)])
@pytest.mark.parametrize('masked', [False, True])
def test_kl_iterations_match_oracle(shape, dtype, masked):
    """The Kullback-Leibler update (grads.py:143-160) through the same tiers: its [N, F] ratio products
    take the narrow tiles on both sides."""
    import torch
    from decomp_amd import _arrays, _hip
    from oracle import nmf as onmf, common
    N, F, K = shape
    y, d0, m = _problem(N, F, K, dtype, masked, seed=7 + N + F + K)
    Yg, Dg = torch.from_numpy(y).cuda(), torch.from_numpy(d0).cuda()
    Mg = None if m is None else torch.from_numpy(m).cuda()
    _arrays.l2_normalize_(Dg, strict=True)
    xg = torch.ones((N, K), device='cuda', dtype=Yg.dtype)
    lib, h = _arrays.lib_handle(Yg)
    sfx = 'f32' if dtype == np.float32 else 'f64'
    ctype = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    fn = getattr(lib, 'dcp_nmf_mu_' + sfx)
    it = ctypes.c_int(0)
    x, d = np.ones((N, K), dtype), common.l2_strict(d0)
    tol_x, tol_d = (2e-4, 2e-4) if dtype == np.float32 else (1e-10, 1e-10)
    for step in range(3):
        _hip.check(h, fn(h, _arrays.ptr(Yg), _arrays.ptr(Mg), _arrays.ptr(xg), _arrays.ptr(Dg), N, F, K,
                         _hip.LIK_KL, ctype(0.0), 2, ctypes.byref(it), None, None), 'dcp_nmf_mu kl')
        x, d, _ = onmf.mu_step(y, x, d, m, 'kl')
        dx = float(np.max(np.abs(xg.cpu().numpy() - x)) / max(1.0, float(np.max(np.abs(x)))))
        dd = float(np.max(np.abs(Dg.cpu().numpy() - d)))
        assert dx <= tol_x and dd <= tol_d, (shape, dtype, masked, step, dx, dd)


@pytest.mark.parametrize('masked', [False, True])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('shape', SHAPES)
def test_mu_iterations_match_oracle(shape, dtype, masked):
    import torch
    from decomp_amd import _arrays, _hip
    from oracle import nmf as onmf, common
    N, F, K = shape
    if masked and K > 128:
        pytest.skip('masked variants: the narrow tiers are what this file adds')
    y, d0, m = _problem(N, F, K, dtype, masked, seed=N + F + K)
    Yg, Dg = torch.from_numpy(y).cuda(), torch.from_numpy(d0).cuda()
    Mg = None if m is None else torch.from_numpy(m).cuda()
    _arrays.l2_normalize_(Dg, strict=True)
    xg = torch.ones((N, K), device='cuda', dtype=Yg.dtype)
    lib, h = _arrays.lib_handle(Yg)
    sfx = 'f32' if dtype == np.float32 else 'f64'
    ctype = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    fn = getattr(lib, 'dcp_nmf_mu_' + sfx)
    it = ctypes.c_int(0)
    x, d = np.ones((N, K), dtype), common.l2_strict(d0)
    tol_r, tol_d = (2e-5, 2e-4) if dtype == np.float32 else (1e-12, 1e-10)
    for step in range(3):
        _hip.check(h, fn(h, _arrays.ptr(Yg), _arrays.ptr(Mg), _arrays.ptr(xg), _arrays.ptr(Dg), N, F, K,
                         _hip.LIK_L2, ctype(0.0), 2, ctypes.byref(it), None, None), 'dcp_nmf_mu')
        x, d, _ = onmf.mu_step(y, x, d, m)
        r_cpu = _residual(y, x, d, m)
        r_hip = _residual(y, xg.cpu().numpy(), Dg.cpu().numpy(), m)
        assert abs(r_hip - r_cpu) <= tol_r * r_cpu, (shape, dtype, masked, step, r_hip, r_cpu)
        dd = float(np.max(np.abs(Dg.cpu().numpy() - d)))
        assert dd <= tol_d, (shape, dtype, masked, step, dd)
