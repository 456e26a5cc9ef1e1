"""CPU: the C-ABI library builds, loads and exports every symbol include/decomp_hip.h
declares (no compute call is made here: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'decomp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dcp_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported_and_bound():
    from decomp_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_hip.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), 'header declares %s but the library does not export it' % n
        assert n in _hip.SIGNATURES, 'no ctypes signature for %s' % n
    for n in _hip.SIGNATURES:
        assert n in names, '%s is bound but not declared in the header' % n


def test_build_info_and_error_paths_without_gpu():
    from decomp_amd import _hip
    lib = _hip.load()
    assert b'gfx950' in lib.dcp_build_info()
    # width of the all-reduced statistics (pure host arithmetic)
    assert lib.dcp_nmf_mu_stats_width(4096, 256, 0, 0) == 4096 + 256
    assert lib.dcp_nmf_mu_stats_width(4096, 256, 0, 1) == 8192
    assert lib.dcp_nmf_mu_stats_width(4096, 256, 1, 0) == 8192
    # null handle -> DCP_ERR_INVALID, never a crash
    assert lib.dcp_set_stream(None, None) == -1
    assert lib.dcp_profile_enable(None, 1) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from decomp_amd import _hip
    monkeypatch.setattr(_hip, '_lib', None)
    monkeypatch.setattr(_hip, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_hip.HipLibraryError):
        _hip.load()
