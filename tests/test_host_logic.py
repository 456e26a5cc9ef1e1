"""CPU: host-side logic of the drop-in API -- validation order and exception types
(reference: decomp/utils/assertion.py, decomp/nmf.py:52-79) -- everything that happens
before the first byte goes to the GPU."""
import numpy as np
import pytest

import decomp_amd
from decomp_amd import nmf
from decomp_amd.utils import assertion, dtype
from decomp_amd.utils.exceptions import (ShapeMismatchError, DtypeMismatchError,
                                         DimInvalidError)


def test_exception_hierarchy():
    for e in (ShapeMismatchError, DtypeMismatchError, DimInvalidError):
        assert issubclass(e, ValueError)


def test_assert_shapes_modes():
    a, b = np.zeros((4, 3)), np.zeros((3, 5))
    assertion.assert_shapes('a', a, 'b', b, axes=1)
    with pytest.raises(ShapeMismatchError):
        assertion.assert_shapes('a', a, 'b', b)
    with pytest.raises(ShapeMismatchError):
        assertion.assert_shapes('a', a, 'b', np.zeros((4, 5)), axes=1)
    assertion.assert_shapes('a', a, 'b', np.zeros((9, 3)), axes=[-1])
    with pytest.raises(ShapeMismatchError):
        assertion.assert_shapes('a', a, 'b', b, axes=[-1])
    assertion.assert_shapes('a', None, 'b', b)
    with pytest.raises(TypeError):
        assertion.assert_shapes('a', a, 'b', b, axes='x')


def test_assert_dtypes_and_ndim():
    f32, f64 = np.zeros(2, np.float32), np.zeros(2, np.float64)
    assertion.assert_dtypes(a=f64, b=f64, c=None)
    with pytest.raises(DtypeMismatchError):
        assertion.assert_dtypes(a=f32, b=f64)
    with pytest.raises(DtypeMismatchError):
        assertion.assert_dtypes(a=np.zeros(2, np.complex128), dtypes='f')
    with pytest.raises(DtypeMismatchError):
        assertion.assert_dtypes(a=np.zeros(2, np.int32))
    with pytest.raises(DimInvalidError):
        assertion.assert_ndim('a', f32, 2)
    assert dtype.float_type(np.dtype(np.complex64)) == np.float32
    assert dtype.float_type(np.dtype(np.float64)) == np.float64
    with pytest.raises(DtypeMismatchError):
        dtype.float_type(np.dtype(np.int32))


def test_nmf_solve_rejects_bad_arguments_before_touching_the_gpu():
    y = np.abs(np.random.RandomState(0).randn(10, 6))
    D = np.abs(np.random.RandomState(1).randn(3, 6))
    with pytest.raises(DtypeMismatchError):          # f32 vs f64
        nmf.solve(y, D.astype(np.float32))
    with pytest.raises(DtypeMismatchError):          # mask must share y's float dtype (nmf.py:57)
        nmf.solve(y, D, mask=np.ones((10, 6), np.int64))
    with pytest.raises(DtypeMismatchError):          # complex is not allowed for NMF
        nmf.solve(y.astype(complex), D.astype(complex))
    with pytest.raises(ShapeMismatchError):          # y and D disagree on channels
        nmf.solve(y, np.abs(np.random.randn(3, 5)))
    with pytest.raises(ShapeMismatchError):          # x vs D
        nmf.solve(y, D, x=np.ones((10, 4)))
    with pytest.raises(ShapeMismatchError):          # mask vs y
        nmf.solve(y, D, mask=np.ones((10, 5)))
    with pytest.raises(DimInvalidError):
        nmf.solve(y[None], D, x=np.ones((1, 10, 3)))


def test_constants_match_reference_surface():
    assert nmf.BATCH_METHODS == ['mu']
    assert nmf.MINIBATCH_METHODS == ['asg-mu', 'gsg-mu', 'asag-mu', 'gsag-mu', 'svrmu',
                                     'svrmu-acc']
    assert hasattr(decomp_amd, 'nmf')


def test_no_cpu_fallback_without_gpu():
    """Valid arguments + no GPU must raise, not compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from decomp_amd._hip import HipLibraryError
    y = np.abs(np.random.RandomState(0).randn(10, 6))
    D = np.abs(np.random.RandomState(1).randn(3, 6))
    with pytest.raises(HipLibraryError):
        nmf.solve(y, D)


def test_product_never_imports_the_oracle():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, 'decomp_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
