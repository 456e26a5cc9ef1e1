"""GPU: the GEMM cores behind every solver (fp32 MFMA, generic fp64) against NumPy
float64, through the C ABI test hook dcp_gemm_*."""
import numpy as np
import pytest

from gpu_util import gemm_hip, gemm_ref, gemm_bound

pytestmark = pytest.mark.gpu


def _operands(form, M, N, K, dtype, rng):
    # asymmetric data: a swapped row/col map or a transposed operand cannot pass
    if form == 0:
        A, B = rng.randn(M, K), rng.randn(N, K)
    elif form == 1:
        A, B = rng.randn(M, K), rng.randn(K, N)
    else:
        A, B = rng.randn(K, M), rng.randn(K, N)
    A = A + np.arange(A.shape[1])[None, :] * 0.01
    B = B - np.arange(B.shape[0])[:, None] * 0.02
    return A.astype(dtype), B.astype(dtype)


SHAPES_ALIGNED = [(256, 256, 64), (512, 384, 4096), (128, 128, 16), (1024, 256, 512)]
SHAPES_RAGGED = [(101, 20, 3), (37, 65, 129), (1, 1, 1), (3, 101, 20), (130, 258, 35),
                 (257, 8, 4096)]


@pytest.mark.parametrize('form', [0, 1, 2])
@pytest.mark.parametrize('tile', list(range(0, 35)))
def test_f32_aligned(form, tile):
    rng = np.random.RandomState(form * 10 + tile)
    for (M, N, K) in SHAPES_ALIGNED:
        A, B = _operands(form, M, N, K, np.float32, rng)
        C = gemm_hip(form, A, B, tile=tile)
        err = np.abs(C - gemm_ref(form, A, B)) / gemm_bound(form, A, B)
        assert err.max() < 2e-5, (form, tile, M, N, K, err.max())


@pytest.mark.parametrize('form', [0, 1, 2])
@pytest.mark.parametrize('tile', list(range(0, 35)))
def test_f32_ragged(form, tile):
    rng = np.random.RandomState(100 + form * 10 + tile)
    for (M, N, K) in SHAPES_RAGGED:
        A, B = _operands(form, M, N, K, np.float32, rng)
        C = gemm_hip(form, A, B, tile=tile)
        err = np.abs(C - gemm_ref(form, A, B)) / (gemm_bound(form, A, B) + 1e-30)
        assert err.max() < 2e-5, (form, tile, M, N, K, err.max())


@pytest.mark.parametrize('form', [0, 1, 2])
@pytest.mark.parametrize('ksplits', [2, 5, 64])
def test_f32_splitk(form, ksplits):
    rng = np.random.RandomState(7 + ksplits)
    for (M, N, K) in [(256, 128, 2048), (130, 70, 1000), (64, 64, 48)]:
        for tile in (1, 2):
            A, B = _operands(form, M, N, K, np.float32, rng)
            C = gemm_hip(form, A, B, ksplits=ksplits, tile=tile)
            err = np.abs(C - gemm_ref(form, A, B)) / gemm_bound(form, A, B)
            assert err.max() < 2e-5, (form, ksplits, tile, M, N, K, err.max())


def test_f32_identity_asymmetric():
    """A = I with an asymmetric B catches a transposed C write (guide, section 3)."""
    n = 128
    B = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 251) - 100.0
    I = np.eye(n, dtype=np.float32)
    assert np.array_equal(gemm_hip(1, I, B, tile=1), B)       # NN: I . B
    assert np.array_equal(gemm_hip(0, I, B, tile=1), B.T)     # NT: I . B^T
    assert np.array_equal(gemm_hip(2, I, B, tile=1), B)       # TN: I^T . B
    assert np.array_equal(gemm_hip(0, B, I, tile=1), B)       # NT: B . I^T


def test_f32_bitwise_reproducible():
    rng = np.random.RandomState(3)
    A, B = _operands(2, 256, 384, 8192, np.float32, rng)
    C1 = gemm_hip(2, A, B, ksplits=16, tile=1)
    C2 = gemm_hip(2, A, B, ksplits=16, tile=1)
    assert np.array_equal(C1, C2)


@pytest.mark.parametrize('form', [0, 1, 2])
def test_f64_generic(form):
    rng = np.random.RandomState(11 + form)
    for (M, N, K) in [(101, 20, 3), (256, 128, 8), (70, 130, 257), (64, 64, 64)]:
        for ks in (1, 3):
            A, B = _operands(form, M, N, K, np.float64, rng)
            C = gemm_hip(form, A, B, ksplits=ks)
            err = np.abs(C - gemm_ref(form, A, B)) / gemm_bound(form, A, B)
            assert err.max() < 1e-14, (form, M, N, K, ks, err.max())


@pytest.mark.parametrize('form', [0, 1, 2])
def test_f64_mfma(form):
    """Outputs of at least 128 x 128 run on the fp64 MFMA core (v_mfma_f64_16x16x4_f64):
    aligned shapes take the vector-load instantiation, ragged ones the bounds-checked one;
    asymmetric operands catch any slip in the (f64-specific) C/D lane map."""
    rng = np.random.RandomState(21 + form)
    for (M, N, K) in SHAPES_ALIGNED + [(130, 258, 35), (257, 129, 4096), (300, 200, 100), (128, 640, 48)]:
        for ks in (1, 3):
            A, B = _operands(form, M, N, K, np.float64, rng)
            C = gemm_hip(form, A, B, ksplits=ks)
            err = np.abs(C - gemm_ref(form, A, B)) / gemm_bound(form, A, B)
            assert err.max() < 1e-14, (form, M, N, K, ks, err.max())
            Cg = gemm_hip(form, A, B, ksplits=ks, tile=2)       # generic VALU core
            assert np.abs(C - Cg).max() <= 1e-13 * np.abs(Cg).max()


def _cplx(rng, *s):
    return (rng.randn(*s) + 1j * rng.randn(*s)).astype(np.complex64)


@pytest.mark.parametrize('form', [0, 1, 2])
def test_c64_on_mfma(form):
    """complex64 products run on the fp32 MFMA core through real extended operands
    (form 0: A B^H, 1: A B, 2: A^H B); aligned, ragged and split-K shapes."""
    rng = np.random.RandomState(40 + form)
    for (M, N, K, ks) in [(256, 256, 128, 1), (128, 64, 512, 4), (101, 20, 3, 1), (37, 65, 129, 3),
                          (512, 128, 256, 1), (64, 96, 1000, 7), (5, 3, 10, 1),
                          # left operand the smaller one: A B takes the planar-rows form (no image of B)
                          (32, 1024, 512, 1), (31, 333, 130, 1), (64, 2048, 64, 1), (16, 512, 2048, 4)]:
        if form == 0:
            A, B = _cplx(rng, M, K), _cplx(rng, N, K)
            ref = A.astype(np.complex128) @ B.astype(np.complex128).conj().T
            bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
        elif form == 1:
            A, B = _cplx(rng, M, K), _cplx(rng, K, N)
            ref = A.astype(np.complex128) @ B.astype(np.complex128)
            bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
        else:
            A, B = _cplx(rng, K, M), _cplx(rng, K, N)
            ref = A.astype(np.complex128).conj().T @ B.astype(np.complex128)
            bound = np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)
        for tile in (0, 1, 2):
            C = gemm_hip(form, A, B, ksplits=ks, tile=tile)
            err = np.abs(C - ref) / bound
            assert err.max() < 2e-5, (form, M, N, K, ks, tile, err.max())


@pytest.mark.parametrize('form', [0, 1, 2])
def test_c128_on_mfma(form):
    """complex128 products whose real-extended output is at least 128 x 128 run on the fp64 MFMA
    core (real extended operands; lane-pair / lane-quad recombination in the epilogue); tile=2
    forces the generic complex core for comparison."""
    rng = np.random.RandomState(50 + form)

    def cplx(*s):
        return rng.randn(*s) + 1j * rng.randn(*s)
    for (M, N, K, ks) in [(256, 256, 128, 1), (128, 64, 512, 4), (130, 70, 35, 1), (64, 96, 1000, 7),
                          (512, 128, 256, 1), (300, 129, 77, 3)]:
        if form == 0:
            A, B = cplx(M, K), cplx(N, K)
            ref = A @ B.conj().T
            bound = np.abs(A) @ np.abs(B).T
        elif form == 1:
            A, B = cplx(M, K), cplx(K, N)
            ref = A @ B
            bound = np.abs(A) @ np.abs(B)
        else:
            A, B = cplx(K, M), cplx(K, N)
            ref = A.conj().T @ B
            bound = np.abs(A).T @ np.abs(B)
        for tile in (0, 2):
            C = gemm_hip(form, A, B, ksplits=ks, tile=tile)
            err = np.abs(C - ref) / bound
            assert err.max() < 1e-14, (form, M, N, K, ks, tile, err.max())


@pytest.mark.parametrize('dtype', [np.float32, np.float64, np.complex64, np.complex128])
def test_random_shapes_auto_dispatch(dtype):
    """Seeded random shapes through the AUTOMATIC tile choice (tile 0), all three forms, with and
    without split-K: every tier of pick_tier / f64_tier, the bounds-checked loaders (clamped 16-byte
    loads on aligned operands, element-wise loads on odd leading dimensions) and the complex paths
    (extended image, planar rows, TN real views)."""
    rng = np.random.RandomState({np.float32: 1, np.float64: 2, np.complex64: 3, np.complex128: 4}[dtype])
    cplx = np.issubdtype(dtype, np.complexfloating)
    tol = 2e-5 if dtype in (np.float32, np.complex64) else 1e-13
    sizes = [1, 3, 8, 20, 31, 32, 33, 50, 64, 65, 100, 127, 128, 129, 200, 250, 256, 260, 300, 500, 512, 777, 1024, 2050]
    for trial in range(60):
        M, N = int(rng.choice(sizes)), int(rng.choice(sizes))
        K = int(rng.choice([1, 5, 16, 33, 64, 100, 256, 511, 1024, 3000]))
        if trial % 7 == 0:
            M = int(rng.choice([4096, 5000, 8192]))     # tall: the big row-tile tiers
        form = trial % 3
        ks = int(rng.choice([1, 1, 2, 5]))

        def mk(*s):
            a = rng.randn(*s)
            if cplx:
                a = a + 1j * rng.randn(*s)
            return a.astype(dtype)
        if form == 0:
            A, B = mk(M, K), mk(N, K)
            ref = A.astype(np.complex128 if cplx else np.float64) @ B.astype(np.complex128 if cplx else np.float64).conj().T
            bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
        elif form == 1:
            A, B = mk(M, K), mk(K, N)
            ref = A.astype(np.complex128 if cplx else np.float64) @ B.astype(np.complex128 if cplx else np.float64)
            bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
        else:
            A, B = mk(K, M), mk(K, N)
            ref = A.astype(np.complex128 if cplx else np.float64).conj().T @ B.astype(np.complex128 if cplx else np.float64)
            bound = np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)
        C = gemm_hip(form, A, B, ksplits=ks, tile=0)
        err = np.abs(C - ref) / (bound + 1e-300)
        assert err.max() < tol, (dtype, form, M, N, K, ks, float(err.max()))


def _tn_plain(on):
    from decomp_amd import _hip
    return _hip.load().dcp_debug_tn_plain(int(on))


def test_tn_pair_schedule_equals_plain_schedule():
    """the pair schedule (TileCfg PIPE = 3) of the reduction-over-samples products rests on hand-placed
    LDS reads and counted waits in inline asm.  The same products -- aligned, ragged, split-K, and the two-segment
    [Y | x] operand with an ODD seam through the NMF statistics -- run with the pair schedule and with the plain
    schedule (dcp_debug_tn_plain) and must agree to rounding level with each other and with float64: a toolchain
    change that breaks the asm ordering shows up here."""
    import ctypes
    import torch
    from decomp_amd import _arrays, _hip
    rng = np.random.RandomState(5)
    prev = _tn_plain(-1)
    try:
        for (M, N, K, ks, tile) in [(256, 4352, 8192, 15, 1), (130, 258, 1035, 1, 1), (512, 1088, 4096, 4, 0),
                                    (512, 4608, 8192, 7, 0), (128, 128, 48, 1, 1), (64, 200, 999, 3, 1)]:
            A, B = _operands(2, M, N, K, np.float32, rng)
            _tn_plain(0)
            C_pair = gemm_hip(2, A, B, ksplits=ks, tile=tile)
            _tn_plain(1)
            C_plain = gemm_hip(2, A, B, ksplits=ks, tile=tile)
            bound = gemm_bound(2, A, B)
            assert (np.abs(C_pair - gemm_ref(2, A, B)) / bound).max() < 2e-5, (M, N, K)
            assert (np.abs(C_pair - C_plain) / bound).max() < 2e-6, (M, N, K)
        # two-segment B = [Y | x] with an odd seam (F = 1201) and ragged rows, through dcp_nmf_mu_stats_f32
        N, F, K = 1003, 1201, 24
        Y = torch.rand((N, F), device='cuda')
        D = torch.rand((K, F), device='cuda') + 0.1
        x = torch.rand((N, K), device='cuda') + 0.1
        lib, h = _arrays.lib_handle(D)
        W = lib.dcp_nmf_mu_stats_width(F, K, 0, 0)
        out = []
        for plain in (0, 1):
            _tn_plain(plain)
            stats = torch.zeros((K, W), device='cuda')
            xo = torch.empty_like(x)
            _hip.check(h, lib.dcp_nmf_mu_stats_f32(h, _arrays.ptr(Y), None, _arrays.ptr(x), _arrays.ptr(xo),
                                                   _arrays.ptr(D), N, F, K, 0, _arrays.ptr(stats)), 'stats')
            out.append((stats.cpu().numpy().astype(np.float64), xo.cpu().numpy().astype(np.float64)))
        assert np.array_equal(out[0][1], out[1][1])                     # the x update does not depend on the knob
        xn, Yn = out[0][1], Y.cpu().numpy().astype(np.float64)
        ref = np.concatenate([xn.T @ Yn, xn.T @ xn], axis=1)
        scale = np.abs(ref).max()
        assert np.abs(out[0][0] - ref).max() / scale < 2e-5
        assert np.abs(out[0][0] - out[1][0]).max() / scale < 2e-6
    finally:
        _tn_plain(prev)
