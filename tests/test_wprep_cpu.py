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


def _decompose(table, n, b):
    """binary search over first_block + the index decomposition both kernels start with"""
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
    dco, dci = (e.dst_cout, e.dst_cin) if e.dst_cout else (e.cout, e.cin)
    k = (((co + e.co_off) * e.kh + r) * e.kw + s) * dci + ci + e.ci_off
    j = (((ci + e.ci_off) * e.kh + (e.kh - 1 - r)) * e.kw + (e.kw - 1 - s)) * dco + co + e.co_off
    return e, src_index, k, j, dco * dci * e.kh * e.kw


def _emulate_kernel(table_bytes, n, total_blocks):
    """prep_weights_kernel (csrc/stp3_wprep.hip), statement by statement"""
    table = (_lib.WprepEntry * n).from_buffer_copy(table_bytes)
    for b in range(total_blocks):
        e, src_index, k, j, dst_total = _decompose(table, n, b)
        if not len(src_index):
            continue
        span = int(src_index.max()) + 1
        src = np.ctypeslib.as_array(ctypes.cast(e.src, ctypes.POINTER(ctypes.c_uint32)), (span,))
        if e.fwd:
            if e.fwd_f32:
                np.ctypeslib.as_array(ctypes.cast(e.fwd, ctypes.POINTER(ctypes.c_uint32)), (dst_total,))[k] = src[src_index]
            else:
                np.ctypeslib.as_array(ctypes.cast(e.fwd, ctypes.POINTER(ctypes.c_uint16)), (dst_total,))[k] = _f2bf(src[src_index])
        if e.flip:
            np.ctypeslib.as_array(ctypes.cast(e.flip, ctypes.POINTER(ctypes.c_uint16)), (dst_total,))[j] = _f2bf(src[src_index])


def _emulate_scatter(table_bytes, n, total_blocks):
    """scatter_weight_grads_kernel: ``src`` is the destination inside the parameter's gradient, ``fwd`` the assembled gradient"""
    table = (_lib.WprepEntry * n).from_buffer_copy(table_bytes)
    for b in range(total_blocks):
        e, dst_index, k, _, whole = _decompose(table, n, b)
        if not len(dst_index):
            continue
        dst = np.ctypeslib.as_array(ctypes.cast(e.src, ctypes.POINTER(ctypes.c_float)), (int(dst_index.max()) + 1,))
        dst[dst_index] = np.ctypeslib.as_array(ctypes.cast(e.fwd, ctypes.POINTER(ctypes.c_float)), (whole,))[k]


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


def test_assembled_weights_and_their_gradient_scatter(monkeypatch):
    """ops.assembled_weight: a weight made of parameter VIEWS at channel offsets of a zero whole -- the two taps of a causal
    (2,3,3) kernel side by side in padded lanes, heads one below the other, a block-diagonal 1x1, the centre tap of a 3x3, the
    leading columns of a projection -- and a depthwise parameter's tap-major float32 copy.  The table rows the host builds, run
    through the kernels' arithmetic, must give what torch's pad / cat / block_diag / slice give (bf16 bit patterns), and the
    scatter rows must put the gradient of the whole back into every parameter's gradient slice (its ``_stp3_grad_view``)."""
    torch.manual_seed(1)
    calls = []

    class FakeLib:
        def stp3_conv2d_prep_weights(self, table, n, total_blocks, stream):
            calls.append('prep')
            _emulate_kernel(ctypes.string_at(table, ctypes.sizeof(_lib.WprepEntry) * n), n, total_blocks)
            return 0

        def stp3_conv2d_scatter_weight_grads(self, table, n, total_blocks, stream):
            calls.append('scatter')
            _emulate_scatter(ctypes.string_at(table, ctypes.sizeof(_lib.WprepEntry) * n), n, total_blocks)
            return 0

    monkeypatch.setattr(_lib, 'lib', lambda: FakeLib())
    monkeypatch.setattr(ops, '_stream', lambda: None)
    sh = ops._WeightShadows('cpu')
    monkeypatch.setitem(ops._SHADOW_TABLES, torch.device('cpu'), sh)
    P = torch.nn.Parameter
    w3d = P(torch.randn(35, 35, 2, 3, 3))                                                   # causal kernel: taps in 40-lane runs
    heads = [P(torch.randn(16, 24, 3, 3).contiguous(memory_format=torch.channels_last)) for _ in range(3)]
    lasts = [P(torch.randn(co, 16, 1, 1)) for co in (2, 1, 4)]
    dil = P(torch.randn(8, 16, 3, 3))
    proj = P(torch.randn(24, 40, 1, 1))
    cases = {
        'taps': ((40, 80, 3, 3), [ops.weight_piece(w3d, w3d.detach().unbind(2)[k], 0, 40 * k) for k in range(2)],
                 lambda: torch.nn.functional.pad(torch.cat([torch.nn.functional.pad(t, (0, 0, 0, 0, 0, 5)) for t in w3d.unbind(2)], dim=1),
                                                 (0, 0, 0, 0, 0, 0, 0, 5))),
        'heads': ((48, 24, 3, 3), [ops.weight_piece(h, h.detach(), 16 * k, 0) for k, h in enumerate(heads)], lambda: torch.cat(list(heads), dim=0)),
        'diag': ((8, 48, 1, 1), [ops.weight_piece(m, m.detach(), co, 16 * k) for k, (m, co) in enumerate(zip(lasts, (0, 2, 3)))],
                 lambda: torch.nn.functional.pad(torch.block_diag(*[m.flatten(1) for m in lasts]), (0, 0, 0, 1))[:, :, None, None]),
        'centre': ((8, 16, 1, 1), [ops.weight_piece(dil, dil.detach()[:, :, 1:2, 1:2])], lambda: dil[:, :, 1:2, 1:2]),
        'columns': ((24, 32, 1, 1), [ops.weight_piece(proj, proj.detach()[:, :32])], lambda: proj[:, :32]),
    }
    ents = {}
    for name, (shape, pieces, ref) in cases.items():
        ent = ents[name] = sh.register_assembled(name, shape, pieces)
        want = ref().detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        want_t = want.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
        assert ent['wb'].permute(0, 2, 3, 1).is_contiguous() and ent['wt'].permute(0, 2, 3, 1).is_contiguous()
        assert torch.equal(ent['wb'].view(torch.int16), want.view(torch.int16)), name
        assert torch.equal(ent['wt'].view(torch.int16), want_t.view(torch.int16)), name
        assert bool(ent['token'].isnan().all()) and tuple(ent['token'].shape) == shape
    dw = P(torch.randn(12, 1, 5, 5))
    taps = sh.register_taps(dw)['taps']
    assert torch.equal(taps, dw.detach().reshape(12, 25).t().contiguous())
    # an optimizer step: ONE launch rewrites leaf shadows, assembled weights and taps
    before = len(calls)
    with torch.no_grad():
        for p in [w3d, dil, dw] + heads:
            p.mul_(0.5)
    sh.refresh()
    assert calls[before:] == ['prep']
    assert torch.equal(taps, dw.detach().reshape(12, 25).t().contiguous())
    assert torch.equal(ents['heads']['wb'].float(), torch.cat(list(heads), dim=0).detach().to(torch.bfloat16).float())
    # the way back: the gradient of each whole, cut into the parameters' gradient slices
    params = [w3d, dil, proj] + heads + lasts
    for p in params:
        p._stp3_grad_view = torch.full_like(p, 7.0)
    for name, ent in ents.items():
        ent['dw'].copy_(torch.randn(ent['shape']))
    sh.scatter(list(ents.values()))
    sh.scatter(list(ents.values()))                           # (the table of a set of weights is built once)
    assert calls.count('scatter') == 2 and len(sh.scatter_tables) == 1
    g = ents['taps']['dw']
    assert torch.equal(w3d._stp3_grad_view, torch.stack([g[:35, 0:35], g[:35, 40:75]], dim=2))
    for k, h in enumerate(heads):
        assert torch.equal(h._stp3_grad_view, ents['heads']['dw'][16 * k:16 * k + 16])
    for k, (m, co) in enumerate(zip(lasts, (0, 2, 3))):
        assert torch.equal(m._stp3_grad_view, ents['diag']['dw'][co:co + m.shape[0], 16 * k:16 * k + 16])
    want = torch.full_like(dil, 7.0)                          # (elements no piece covers are the caller's to zero)
    want[:, :, 1:2, 1:2] = ents['centre']['dw']
    assert torch.equal(dil._stp3_grad_view, want)
    want = torch.full_like(proj, 7.0)
    want[:, :32] = ents['columns']['dw']
    assert torch.equal(proj._stp3_grad_view, want)
