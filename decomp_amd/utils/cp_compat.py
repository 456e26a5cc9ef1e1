"""Array-module dispatch -- drop-in for ``decomp.utils.cp_compat`` (cp_compat.py:4-24).

The reference threads an array module ``xp`` (numpy or cupy) through every solver.  Here
the device arrays are torch CUDA tensors (plumbing only: memory, streams), so
``numpy_or_cupy`` names the module that CARRIES device arrays and ``has_cupy`` keeps its
meaning "a GPU array type is available".  ``get_array_module`` returns that module object
for device arrays and ``numpy`` for NumPy arrays, and raises the reference's ``TypeError``
when the arguments mix kinds (cp_compat.py:11-13).
"""
import numpy

from .. import _arrays

try:
    import torch as _torch
    numpy_or_cupy = _torch
    has_cupy = True
except ImportError:  # pragma: no cover
    _torch = None
    numpy_or_cupy = numpy
    has_cupy = False


def get_array_module(*arrays):
    kind = _arrays.get_array_module(*arrays)
    return _torch if kind == 'torch' else numpy
