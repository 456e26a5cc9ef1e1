#!/usr/bin/env python3
"""FETCH_SIZE calibration for the GEMM loaders' access patterns (run under rocprofv3 --pmc FETCH_SIZE):
reads a 2 GiB float matrix (larger than L2 + Infinity Cache) once per pattern."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from decomp_amd import _arrays, _hip  # noqa: E402

rows, cols = 131072, 4096          # 2 GiB
Y = torch.rand((rows, cols), device='cuda')
out = torch.zeros((4,), device='cuda')
lib, h = _arrays.lib_handle(Y)
for pattern in (0, 1, 0, 1):
    _hip.check(h, lib.dcp_calib_read_f32(h, _arrays.ptr(Y), rows, cols, pattern, _arrays.ptr(out)), 'calib')
torch.cuda.synchronize()
print('known bytes per launch: %d' % (rows * cols * 4))
