"""CPU: arithmetic and host side of the squeeze-and-excitation MLP kernels (STP3_SE_MLP, csrc/stp3_se_mlp.hip).

The three kernels are transliterated to numpy (same loops: per-sample chain, weight gradients reduced over the
samples in ascending order, float32) and plugged in for the library; ``_SeBlock`` then runs on CPU tensors with the
pooling / scaling passes also emulated.  Forward and all five gradients must agree with torch autograd through the
plain statement  y = x * sigmoid(W2 swish(W1 mean(x) + b1) + b2)  (float32: rtol 1e-4, atol 1e-5)."""
import ctypes

import numpy as np
import torch

from stp3_amd import _lib, ops, ops_fused


def _arr(ptr, shape, dtype=np.float32):
    n = int(np.prod(shape))
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dtype))), (n,)).reshape(shape)


def _sig(z):
    return (1.0 / (1.0 + np.exp(-z.astype(np.float32)))).astype(np.float32)


class FakeLib:
    """stp3_se_pool / stp3_se_scale / stp3_se_mlp_* on host memory (channels-last float32 activations)."""

    def stp3_se_workspace_bytes(self, dims, out):
        out._obj.value = 1024
        return 0

    def stp3_se_pool(self, dims, x, dy, ws, ws_bytes, out, stream):
        d = dims._obj
        assert d.dtype == _lib.DTYPE_F32 and d.ld == d.C
        xs = _arr(x, (d.N, d.rows, d.C))
        o = _arr(out, (d.N, d.C))
        o[:] = (xs * _arr(dy, (d.N, d.rows, d.C))).sum(1) if dy else xs.sum(1)
        return 0

    def stp3_se_scale(self, dims, x, gate, add, y, stream):
        d = dims._obj
        out = _arr(y, (d.N, d.rows, d.C))
        out[:] = _arr(x, (d.N, d.rows, d.C)) * _arr(gate, (d.N, 1, d.C))
        if add:
            out += _arr(add, (d.N, 1, d.C))
        return 0

    def stp3_se_mlp_fwd(self, dims, pooled_sum, w1, b1, w2, b2, z1, gate, stream):
        d = dims._obj
        f32 = np.float32
        W1, W2 = _arr(w1, (d.S, d.C)), _arr(w2, (d.C, d.S))
        for n in range(d.N):
            p = _arr(pooled_sum, (d.N, d.C))[n] * f32(d.inv_rows)
            z = (W1 * p).sum(1, dtype=f32) + _arr(b1, (d.S,))
            _arr(z1, (d.N, d.S))[n] = z
            h = z * _sig(z)
            _arr(gate, (d.N, d.C))[n] = _sig((W2 * h).sum(1, dtype=f32) + _arr(b2, (d.C,)))
        return 0

    def stp3_se_mlp_bwd(self, dims, dgate, gate, pooled_sum, z1, w1, w2, dz2, dz1, dpooled, dw1, db1, dw2, db2, stream):
        d = dims._obj
        f32 = np.float32
        W1, W2 = _arr(w1, (d.S, d.C)), _arr(w2, (d.C, d.S))
        G, Z = _arr(gate, (d.N, d.C)), _arr(z1, (d.N, d.S))
        D2, D1 = _arr(dz2, (d.N, d.C)), _arr(dz1, (d.N, d.S))
        for n in range(d.N):                                     # se_mlp_bwd_sample_kernel
            D2[n] = _arr(dgate, (d.N, d.C))[n] * G[n] * (f32(1) - G[n])
            dh = (D2[n][:, None] * W2).sum(0, dtype=f32)
            sg = _sig(Z[n])
            D1[n] = dh * sg * (f32(1) + Z[n] * (f32(1) - sg))
            _arr(dpooled, (d.N, d.C))[n] = (D1[n][:, None] * W1).sum(0, dtype=f32) * f32(d.inv_rows)
        H = Z * _sig(Z)                                          # se_mlp_bwd_weight_kernel
        P = _arr(pooled_sum, (d.N, d.C)) * f32(d.inv_rows)
        a2 = np.zeros((d.C, d.S), f32)
        a1 = np.zeros((d.S, d.C), f32)
        sb, s1 = np.zeros(d.C, f32), np.zeros(d.S, f32)
        for n in range(d.N):
            a2 += D2[n][:, None] * H[n][None, :]
            a1 += D1[n][:, None] * P[n][None, :]
            sb += D2[n]
            s1 += D1[n]
        _arr(dw2, (d.C, d.S))[:], _arr(dw1, (d.S, d.C))[:] = a2, a1
        _arr(db2, (d.C,))[:], _arr(db1, (d.S,))[:] = sb, s1
        return 0


def test_se_block_with_mlp_kernels_matches_autograd(monkeypatch):
    monkeypatch.setattr(_lib, 'lib', lambda: FakeLib())
    monkeypatch.setattr(ops_fused, '_SE_MLP', True)
    monkeypatch.setattr(ops, '_need_gpu', lambda *a: None)
    monkeypatch.setattr(ops, '_stream_handle', lambda: 0)
    torch.manual_seed(0)
    for n, c, s, hh, ww in [(5, 48, 12, 6, 7), (3, 200, 9, 4, 4), (2, 16, 4, 1, 3)]:
        x = torch.randn(n, c, hh, ww).contiguous(memory_format=torch.channels_last).requires_grad_()
        w1 = (torch.randn(s, c, 1, 1) * 0.3).requires_grad_()
        b1 = torch.randn(s).requires_grad_()
        w2 = (torch.randn(c, s, 1, 1) * 0.3).requires_grad_()
        b2 = torch.randn(c).requires_grad_()
        y = ops_fused._SeBlock.apply(x, w1, b1, w2, b2)
        gy = torch.randn_like(y)
        y.backward(gy)
        got = [y.detach()] + [t.grad.clone() for t in (x, w1, b1, w2, b2)]
        for t in (x, w1, b1, w2, b2):
            t.grad = None
        pooled = x.mean(dim=(2, 3))
        z = torch.nn.functional.silu(pooled @ w1.flatten(1).t() + b1)
        gate = torch.sigmoid(z @ w2.flatten(1).t() + b2)
        ref = x * gate[:, :, None, None]
        ref.backward(gy)
        want = [ref.detach()] + [t.grad for t in (x, w1, b1, w2, b2)]
        for a, b in zip(got, want):
            assert a.shape == b.shape
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
