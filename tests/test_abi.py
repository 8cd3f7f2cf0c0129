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
    assert lib.hn_version() == built_lib.HN_VERSION == 121


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


def test_host_side_code_is_clean_under_address_and_ub_sanitizers():
    """SURVEY 5 (sanitizers on the CPU side only: GPU ASan is not available on the pool): the pure-host pieces of the library
    -- the layout planner of the f16x3 weight-stream packer (hn_pack2.hip compiled host-only) and the pose-chain header through
    the oracle's double-precision entry points -- built with -fsanitize=address,undefined and run (tests/san/)."""
    import subprocess
    san = os.path.join(ROOT, 'tests', 'san')
    subprocess.check_call(['make', '-C', san, '-s'], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    for exe, ok in (('pose_chain_san', 'pose chain: ok'), ('pack_layout_san', 'pack layout: ok')):
        r = subprocess.run([os.path.join(san, '_build', exe)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and ok in r.stdout and 'Sanitizer' not in r.stderr and 'runtime error' not in r.stderr, \
            '%s: rc %d\n%s\n%s' % (exe, r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_field_kernels_wait_for_no_tape_load_in_front_of_their_mfmas(built_lib):
    """A static check of the built field kernels' ISA (tools/isa_early_waits.py): in no weight-chunk step may a vector load issued in
    that step be waited for before a quarter of the step's MFMAs have run -- that is a memory round trip in front of the matrix work
    (round 4 found one per step of the adjoint kernels' forward-direction sweep that way: a multiply placed right behind the tape
    load).  Known and accepted: the 8 steps of the second reverse sweep's first layer in the adjoint modes (2, 4, 5), whose `pre` adds
    the W8-row term to the loaded tile."""
    import sys
    build = os.path.join(ROOT, 'ho-nerf_amd', 'csrc', 'build')
    objs = [os.path.join(build, f) for f in ('hn_field2_hand.o', 'hn_field2_hand_adj.o', 'hn_field2_obj.o')]
    if not all(os.path.exists(o) for o in objs) or not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('object files of the field kernels / llvm-objdump not present')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import isa_early_waits
    seen = 0
    for o in objs:
        for name, (nseg, bad) in isa_early_waits.scan(o, 'k_field2').items():
            adjoint = name.endswith('<2>') or name.endswith('<4>') or name.endswith('<5>')
            assert len(bad) <= (9 if adjoint else 0), (name, bad)
            assert nseg > 50, (name, nseg)
            seen += 1
    assert seen >= 9
