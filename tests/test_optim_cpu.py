"""CPU: FlatAdam.clip_and_step -- the default composition (clip_grad_norm_ + step) and the host side + arithmetic of
the fused path (STP3FUSED_ADAM, stp3_optim_clip_adam).

The three kernels of csrc/stp3_optim.hip are transliterated to numpy (same work split: 4096-element blocks found by
binary search over first_block, per-block partial sums, one preparing block, the update) and run against the SAME
bucket table the host code would upload; parameters, moments, clipped gradients, step counter and the returned norm
must agree with the torch-operator path to float32 rounding (the kernel uses fused multiply-adds)."""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from stp3_amd import _lib, parallel
from stp3_amd.parallel import FlatAdam, GradientBuckets


def _arr(ptr, n, dtype=np.float32):
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dtype))), (n,))


def _emulate(table_ptr, n, total_blocks, max_norm, lr, b1, b2, eps, wd, state_ptr, ws_ptr):
    table = (_lib.OptimBucket * n).from_buffer_copy(ctypes.string_at(table_ptr, ctypes.sizeof(_lib.OptimBucket) * n))
    f32 = np.float32

    def bucket_of(b):
        lo, hi = 0, n - 1
        while lo < hi:
            mid = (lo + hi + 1) >> 1
            if table[mid].first_block <= b:
                lo = mid
            else:
                hi = mid - 1
        return table[lo]

    partial = _arr(ws_ptr, total_blocks)
    for b in range(total_blocks):
        e = bucket_of(b)
        base = (b - e.first_block) * 4096
        g = _arr(e.grad, e.numel)[base:base + 4096].astype(np.float64)
        partial[b] = f32((g * g).sum())
    state = _arr(state_ptr, 5)
    total = f32(np.sqrt(partial.astype(np.float64).sum()))
    t = f32(state[0] + 1)
    state[0] = t
    state[1] = min(f32(max_norm) / (total + f32(1e-6)), f32(1)) if max_norm > 0 else f32(1)
    state[2] = f32(lr) / (f32(1) - f32(b1) ** t)
    state[3] = np.sqrt(f32(1) - f32(b2) ** t)
    state[4] = total
    for b in range(total_blocks):
        e = bucket_of(b)
        sl = slice((b - e.first_block) * 4096, (b - e.first_block) * 4096 + 4096)
        grad, param = _arr(e.grad, e.numel), _arr(e.param, e.numel)
        m, v = _arr(e.exp_avg, e.numel), _arr(e.exp_avg_sq, e.numel)
        gc = grad[sl] * state[1]
        g = (gc + f32(wd) * param[sl]) if wd else gc
        m_new = m[sl] + (f32(1) - f32(b1)) * (g - m[sl])
        v_new = v[sl] * f32(b2) + (f32(1) - f32(b2)) * g * g
        denom = np.sqrt(v_new) / state[3] + f32(eps)
        grad[sl], m[sl], v[sl] = gc, m_new, v_new
        param[sl] = param[sl] - (m_new / denom) * state[2]


def _model():
    torch.manual_seed(5)
    return nn.Sequential(nn.Conv2d(3, 16, 3, padding=1), nn.BatchNorm2d(16), nn.ReLU(), nn.Conv2d(16, 8, 3, padding=1),
                         nn.Flatten(), nn.Linear(8 * 6 * 6, 300), nn.ReLU(), nn.Linear(300, 5))


def test_fused_clip_adam_matches_the_torch_operator_path(monkeypatch):
    """Every step starts both optimizers from the SAME state (parameters, gradients, moments, step counter), so the
    comparison is one update at a time and not a trajectory (Adam turns 1-ulp differences near g = 0 into +-lr)."""
    ref_model, fus_model = _model(), _model()
    ref_b = GradientBuckets(ref_model, bucket_bytes=20000)
    fus_b = GradientBuckets(fus_model, bucket_bytes=20000, gather=False)    # filled by copies below, no backward
    assert len(ref_b.buckets) >= 3 and max(f.numel() for f, _ in ref_b.buckets) > 4096   # several blocks per bucket
    ref_opt = FlatAdam(ref_b, lr=1e-2, weight_decay=1e-3)
    fus_opt = FlatAdam(fus_b, lr=1e-2, weight_decay=1e-3)
    calls = []

    class FakeLib:
        def stp3_optim_workspace_bytes(self, total_blocks, out):
            out._obj.value = max(total_blocks, 1) * 4
            return 0

        def stp3_optim_clip_adam(self, table, n, total_blocks, max_norm, lr, b1, b2, eps, wd, state, ws, ws_bytes,
                                 stream):
            assert ws_bytes >= total_blocks * 4
            calls.append((n, total_blocks))
            _emulate(table, n, total_blocks, max_norm, lr, b1, b2, eps, wd, state, ws)
            return 0

    from stp3_amd import ops
    g = torch.Generator().manual_seed(2)
    for it in range(4):
        x = torch.randn(8, 3, 6, 6, generator=g)
        ref_b.zero_grad()
        ref_model(x).square().mean().backward()
        ref_b.finish()
        with torch.no_grad():                                     # same state on both sides
            for (fg, _), (rg, _) in zip(fus_b.buckets, ref_b.buckets):
                fg.copy_(rg)
            for k in range(len(ref_b.buckets)):
                fus_b.flat_params[k].copy_(ref_b.flat_params[k])
                fus_opt.exp_avg[k].copy_(ref_opt.exp_avg[k])
                fus_opt.exp_avg_sq[k].copy_(ref_opt.exp_avg_sq[k])
            fus_opt.step_t.copy_(ref_opt.step_t)
        max_norm = 0.05 if it != 1 else 1e9                       # step 1: the clip does not bind
        n_ref = float(ref_opt.clip_and_step(max_norm))
        with monkeypatch.context() as mp:
            mp.setattr(_lib, 'lib', lambda: FakeLib())
            mp.setattr(parallel, 'FUSED_ADAM', True)
            mp.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))
            mp.setattr(ops, '_stream', lambda: None)
            n_fus = float(fus_opt.clip_and_step(max_norm))
        assert abs(n_fus - n_ref) <= 1e-5 * n_ref
        assert fus_opt.step_count == ref_opt.step_count == it + 1
        for k in range(len(ref_b.buckets)):
            torch.testing.assert_close(fus_b.buckets[k][0], ref_b.buckets[k][0], rtol=5e-5, atol=1e-10)   # clipped g
            torch.testing.assert_close(fus_opt.exp_avg[k], ref_opt.exp_avg[k], rtol=5e-5, atol=1e-10)
            torch.testing.assert_close(fus_opt.exp_avg_sq[k], ref_opt.exp_avg_sq[k], rtol=5e-5, atol=1e-12)
            torch.testing.assert_close(fus_b.flat_params[k], ref_b.flat_params[k], rtol=1e-5, atol=1e-5)   # lr * 1e-3: g ~ 0
    assert len(calls) == 4 and calls[0][0] == len(fus_b.buckets)
    for pr, pf in zip(ref_model.parameters(), fus_model.parameters()):     # the parameters ARE the flat buffers
        torch.testing.assert_close(pf, pr, rtol=1e-5, atol=1e-5)
