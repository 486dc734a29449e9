"""CPU: the weight-shadow table of STP3_WEIGHT_PREP (ops._WeightShadows) and the index arithmetic of
stp3_conv2d_prep_weights.

The kernel itself needs a GPU (tests/test_conv_v2_gpu.py runs it under STP3_EXPERIMENTAL=1); here its body is
transliterated statement by statement (binary search over first_block, index decomposition, flipped destination,
round-to-nearest-even) and executed on the host against the SAME table bytes the host code would upload, for
channels-last and NCHW-contiguous masters.  The result must equal what the default path builds with torch:
``w.to(bfloat16).contiguous(channels_last)`` and ``.flip(2, 3).transpose(0, 1).contiguous(channels_last)``.
"""
import ctypes

import numpy as np
import torch

from stp3_amd import _lib, ops


def _f2bf(u):
    """round to nearest even on the raw float32 bits (uint32 array) -> uint16"""
    u = u.astype(np.uint64)
    nan = (u & 0x7fffffff) > 0x7f800000
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return np.where(nan, ((u >> 16) | 0x40).astype(np.uint16), r)


def _emulate_kernel(table_bytes, n, total_blocks):
    table = (_lib.WprepEntry * n).from_buffer_copy(table_bytes)
    for b in range(total_blocks):
        lo, hi = 0, n - 1
        while lo < hi:
            mid = (lo + hi + 1) >> 1
            if table[mid].first_block <= b:
                lo = mid
            else:
                hi = mid - 1
        e = table[lo]
        total = e.cout * e.cin * e.kh * e.kw
        i = (b - e.first_block) * 256 + np.arange(256)
        i = i[i < total]
        ci = i % e.cin
        t = i // e.cin
        s = t % e.kw
        t = t // e.kw
        r = t % e.kh
        co = t // e.kh
        src_index = co * e.stride_co + ci * e.stride_ci + r * e.stride_kh + s * e.stride_kw
        span = int(src_index.max()) + 1
        src = np.ctypeslib.as_array(ctypes.cast(e.src, ctypes.POINTER(ctypes.c_uint32)), (span,))
        h = _f2bf(src[src_index])
        fwd = np.ctypeslib.as_array(ctypes.cast(e.fwd, ctypes.POINTER(ctypes.c_uint16)), (total,))
        fwd[i] = h
        flip = np.ctypeslib.as_array(ctypes.cast(e.flip, ctypes.POINTER(ctypes.c_uint16)), (total,))
        j = ((ci * e.kh + (e.kh - 1 - r)) * e.kw + (e.kw - 1 - s)) * e.cout + co
        flip[j] = h


def test_shadow_table_and_kernel_arithmetic(monkeypatch):
    torch.manual_seed(0)
    shapes = [(8, 16, 3, 3), (24, 8, 1, 1), (5, 40, 7, 7), (64, 64, 3, 3), (3, 8, 5, 5)]
    weights = []
    for k, shp in enumerate(shapes):
        w = torch.randn(*shp) * 10.0 ** (k - 2)
        if k % 2 == 0:
            w = w.contiguous(memory_format=torch.channels_last)
        weights.append(torch.nn.Parameter(w))
    weights[1].data[0, 0, 0, 0] = float('nan')
    weights[1].data[2, 3, 0, 0] = float('inf')
    weights[3].data[1, 1, 1, 1] = 3.0e38           # rounds up to infinity in bf16
    launches = []

    class FakeLib:
        def stp3_conv2d_prep_weights(self, table, n, total_blocks, stream):
            launches.append((n, total_blocks))
            size = ctypes.sizeof(_lib.WprepEntry) * n
            _emulate_kernel(ctypes.string_at(table, size), n, total_blocks)
            return 0

    monkeypatch.setattr(_lib, 'lib', lambda: FakeLib())
    monkeypatch.setattr(ops, '_stream', lambda: None)
    sh = ops._WeightShadows('cpu')
    for w in weights:
        assert sh.lookup(w) is None
        sh.register(w)
    assert launches[-1] == (len(weights), sum((w.numel() + 255) // 256 for w in weights))

    def check():
        for w in weights:
            ent = sh.lookup(w)
            ref_b = w.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            ref_t = ref_b.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
            assert ent['wb'].shape == ref_b.shape and ent['wt'].shape == ref_t.shape
            # the kernels take the raw pointer: memory must be [Cout][KH][KW][Cin] / [Cin][KH][KW][Cout]
            assert ent['wb'].permute(0, 2, 3, 1).is_contiguous() and ent['wt'].permute(0, 2, 3, 1).is_contiguous()
            assert ref_b.permute(0, 2, 3, 1).is_contiguous() and ref_t.permute(0, 2, 3, 1).is_contiguous()
            for got, ref in ((ent['wb'], ref_b), (ent['wt'], ref_t)):
                same = (got.view(torch.int16) == ref.view(torch.int16)) | (got.isnan() & ref.isnan())
                assert bool(same.all())                      # bit patterns; a NaN stays a NaN (payload is free)

    check()
    n = len(launches)
    with torch.no_grad():                      # an optimizer writing through other storage: explicit refresh
        for w in weights:
            w.data.mul_(1.5)
    sh.refresh()
    assert len(launches) == n + 1
    check()
    with torch.no_grad():                      # an in-place update of the parameter itself is noticed on lookup
        weights[0].add_(1.0)
    assert sh.lookup(weights[0]) is not None and len(launches) == n + 2
    check()
    del weights[2]                             # a parameter that went away drops out of the table
    import gc
    gc.collect()
    sh.refresh()
    assert launches[-1][0] == len(weights)
    check()
