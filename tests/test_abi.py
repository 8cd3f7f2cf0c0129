"""CPU-side checks of the C ABI: the library builds, loads, and exports exactly the
symbols include/honerf.h declares (no compute call is made: no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'honerf.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(hn_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def built_lib():
    from honerf_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return lib


def test_header_lists_functions():
    names = header_functions()
    assert 'hn_render_single' in names and 'hn_render_dual' in names and 'hn_field_create' in names
    assert len(names) >= 20


def test_library_exports_every_declared_symbol(built_lib):
    cdll = ctypes.CDLL(built_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(cdll, name), 'libhonerf.so does not export %s' % name


def test_binding_covers_header(built_lib):
    assert sorted(built_lib.SIGNATURES) == header_functions()
    lib = built_lib.load()
    assert lib.hn_version() == built_lib.HN_VERSION == 108


def test_missing_library_fails_loudly(monkeypatch):
    from honerf_amd import lib
    monkeypatch.setattr(lib, '_lib', None)
    monkeypatch.setattr(lib, 'LIB_PATH', '/nonexistent/libhonerf.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'ho-nerf_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f
