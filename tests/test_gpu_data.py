"""Minibatch containers on the GPU: the reference's container tests (tests/test_utils.py:12-226)
re-expressed for decomp_amd.utils.data, and the out-of-core path (host y / x streamed through
AsyncMinibatchData with a device D, nmf.py:93-103, dictionary_learning.py:87-97) against the
in-core path, whose results are pinned by the golden vectors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _np(a):
    return a.cpu().numpy() if hasattr(a, 'cpu') else np.asarray(a)


def _make(kind, array, minibatch, shuffle_index=None, **kw):
    import torch
    from decomp_amd.utils.data import MinibatchData, AsyncMinibatchData
    if kind == 'incore':
        return MinibatchData(torch.from_numpy(array.copy()).cuda(), minibatch, shuffle_index)
    return AsyncMinibatchData(array.copy(), minibatch, shuffle_index=shuffle_index, **kw)


CONFIGS = {
    'incore': ('incore', (100, 20), 11, {}),
    'incore_even': ('incore', (100, 20), 10, {}),
    'async': ('async', (1000, 20), 100, {}),                           # test_utils.py:177-182
    'async_wo_stream': ('async', (1000, 20), 100, {'use_stream': False}),   # :186-193
    'async_small': ('async', (20, 20), 12, {}),                         # :211-216
    'async_2para': ('async', (1000, 20), 100, {'n_parallel': 2}),       # :220-227
    'async_ragged': ('async', (1003, 7), 100, {'n_parallel': 4}),
}


@pytest.mark.parametrize('cfg', sorted(CONFIGS))
@pytest.mark.parametrize('shuffled', [False, True])
def test_read_loop_write_shuffle(cfg, shuffled):
    kind, shape, mb, kw = CONFIGS[cfg]
    rng = np.random.RandomState(0)
    array = rng.randn(*shape)
    idx = None
    if shuffled:
        idx = np.arange(shape[0])
        rng.shuffle(idx)
    data = _make(kind, array, mb, idx, **kw)
    data2 = _make(kind, array, mb, idx, **kw)
    order = array if idx is None else array[idx]
    n_loop = shape[0] // mb
    assert data.n_loop == n_loop
    # test_read: .array is the original order, minibatches are consecutive row blocks
    assert np.array_equal(_np(data.array), array)
    for _ in range(3):                      # loops can be repeated, alone and zipped
        count = 0
        for i, arr in enumerate(data):
            assert np.array_equal(_np(arr), order[i * mb:(i + 1) * mb]), (cfg, i)
            count += 1
        assert count == n_loop
        count = 0
        for i, (arr, arr2) in enumerate(zip(data, data2)):
            assert np.array_equal(_np(arr), order[i * mb:(i + 1) * mb])
            assert np.array_equal(_np(arr2), order[i * mb:(i + 1) * mb])
            count += 1
        assert count == n_loop
    assert np.array_equal(_np(data.array), array)
    # test_shuffle: cumulative permutation, .array still restores the original order
    for _ in range(2):
        idx2 = np.arange(shape[0])
        rng.shuffle(idx2)
        data.shuffle(idx2)
        order = order[idx2]
        assert np.array_equal(_np(data.array), array)
        for i, arr in enumerate(data):
            assert np.array_equal(_np(arr), order[i * mb:(i + 1) * mb])
    # test_write: in-place changes of the yielded blocks persist (incl. the LAST block, which a
    # zip() that stops on another container never advances past)
    for i, (arr, _) in enumerate(zip(data, data2)):
        arr[...] = float(i)
    assert not np.array_equal(_np(data.array), array)
    count = 0
    for i, arr in enumerate(data):
        assert np.array_equal(_np(arr), np.full(arr.shape, float(i))), (cfg, i)
        count += 1
    assert count == n_loop
    tail = _np(data.array)
    restored = np.empty_like(tail)
    restored[...] = tail
    # rows that never belong to a minibatch (N % minibatch tail of this permutation) are untouched
    cur = data.restore_index
    for pos in range(n_loop * mb, shape[0]):
        assert np.array_equal(restored[cur[pos]], array[cur[pos]])


def test_async_readonly_does_not_write_back_and_errors():
    from decomp_amd.utils.data import AsyncMinibatchData
    from decomp_amd.utils.exceptions import ShapeMismatchError
    a = np.arange(600, dtype=np.float32).reshape(60, 10)
    d = AsyncMinibatchData(a, 16, needs_update=False)
    for arr in d:
        arr.zero_()
    assert np.array_equal(d.array, a)
    with pytest.raises(ValueError):
        AsyncMinibatchData(a, 61)
    with pytest.raises(ShapeMismatchError):
        d.shuffle(np.arange(59))


def _nmf_problem(N=384, F=48, K=6, seed=0):
    rng = np.random.RandomState(seed)
    xt = np.maximum(rng.randn(N, K), 0)
    Dt = np.maximum(rng.randn(K, F), 0)
    y = (xt @ Dt + 0.1 * np.abs(rng.randn(N, F))).astype(np.float32)
    D0 = np.maximum(Dt + 0.3 * rng.randn(K, F), 0.1).astype(np.float32)
    x0 = np.abs(rng.randn(N, K)).astype(np.float32) + 0.1
    mask = (rng.uniform(size=(N, F)) >= 0.2).astype(np.float32)
    return y, D0, x0, mask


@pytest.mark.parametrize('method', ['asg-mu', 'gsag-mu', 'svrmu', 'svrmu-acc'])
@pytest.mark.parametrize('masked', [False, True])
def test_nmf_streamed_equals_incore(method, masked):
    """Host y / x / mask + device D (the reference's out-of-core calling convention) must give
    bit-identical results to the in-core run; x comes back as a NumPy array."""
    import torch
    import decomp_amd
    y, D0, x0, mask = _nmf_problem()
    m = mask if masked else None
    kw = dict(tol=1e-9, minibatch=64, maxiter=4, method=method, random_seed=1)
    it0, D_in, x_in = decomp_amd.nmf.solve(y, D0.copy(), x0.copy(), mask=m, **kw)
    it1, D_st, x_st = decomp_amd.nmf.solve(y, torch.from_numpy(D0).cuda(), x0.copy(), mask=m, **kw)
    assert it0 == it1
    assert isinstance(x_st, np.ndarray) and torch.is_tensor(D_st)
    assert np.array_equal(D_st.cpu().numpy(), D_in)
    assert np.array_equal(x_st, x_in)


@pytest.mark.parametrize('masked', [False, True])
@pytest.mark.parametrize('dt', ['float32', 'complex64'])
def test_dictionary_learning_streamed_equals_incore(masked, dt):
    import torch
    import decomp_amd
    rng = np.random.RandomState(4)
    N, F, K = 320, 40, 12
    cplx = dt == 'complex64'

    def randn(*s):
        return (rng.randn(*s) + 1j * rng.randn(*s)) if cplx else rng.randn(*s)
    Dt = randn(K, F)
    xt = 3.0 * randn(N, K) * (rng.uniform(size=(N, K)) < 0.2)
    y = (xt @ Dt + 0.1 * randn(N, F)).astype(dt)
    D0 = (Dt + 0.2 * randn(K, F)).astype(dt)
    x0 = np.zeros((N, K), dtype=dt)
    mask = np.rint(rng.uniform(0.4, 1.0, size=(N, F))).astype(np.float32) if masked else None
    kw = dict(tol=0.0, minibatch=64, maxiter=3, lasso_method='ista', lasso_iter=10,
              lasso_tol=1e-5, random_seed=0, mask=mask)
    it0, D_in, x_in = decomp_amd.dictionary_learning.solve(y, D0.copy(), 0.01, x0.copy(), **kw)
    it1, D_st, x_st = decomp_amd.dictionary_learning.solve(y, torch.from_numpy(D0).cuda(), 0.01,
                                                           x0.copy(), **kw)
    assert it0 == it1
    assert isinstance(x_st, np.ndarray)
    assert np.array_equal(D_st.cpu().numpy(), D_in)
    assert np.array_equal(x_st, x_in)
    assert np.count_nonzero(x_st) > 0
