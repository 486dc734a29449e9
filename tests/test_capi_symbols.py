"""CPU: libstp3hip.so loads and exports exactly what include/stp3_hip.h declares."""
import ctypes
import os
import re

from stp3_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'stp3_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(stp3_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.SIGNATURES)


def test_library_exports_every_symbol():
    lib = _lib.lib()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.stp3_version().decode().startswith('stp3hip')


def test_argument_validation_without_gpu():
    lib = _lib.lib()
    nbytes = ctypes.c_size_t()
    bad = _lib.LiftDims(0, 3, 6, 48, 28, 60, 64, 200, 200, 1)
    assert lib.stp3_lift_plan_bytes(ctypes.byref(bad), ctypes.byref(nbytes)) == -10001
    ok = _lib.LiftDims(4, 3, 6, 48, 28, 60, 64, 200, 200, 1)
    assert lib.stp3_lift_plan_bytes(ctypes.byref(ok), ctypes.byref(nbytes)) == 0
    assert nbytes.value >= 12 * (40001 + 483840) * 4
    assert lib.stp3_lift_plan_bytes(None, ctypes.byref(nbytes)) == -10001


def test_family_rooflines_count_every_variant_of_an_entry_point():
    """bench.py's per-family rooflines (stp3_amd/profiling.py) time the C-ABI calls listed in ``WORK``: a variant of a listed
    entry (``*_parts``, ``*_partials``, ``*_bn``, ``*_reduce_batch``: the same kernels with a reduction moved) that is missing
    there silently drops its time from the family -- the family would look faster than it is."""
    from stp3_amd import profiling
    listed = set(profiling.WORK)
    for name in _declared():
        for suffix in ('_parts', '_partials', '_bn', '_reduce_batch', '_oihw'):
            if name.endswith(suffix) and name[:-len(suffix)] in listed:
                assert name in listed, name
    assert {'stp3_conv2d_wgrad_partials', 'stp3_conv2d_wgrad_reduce_batch', 'stp3_dwconv2d_fwd_stats_bn'} <= listed
