"""Host side of csrc/stp3_loss.hip: the training losses of the perception path (stp3/losses.py) and the nearest label
warp (stp3/utils/geometry.py:196-238) as operators on the C ABI.  GPU float32 / bf16 tensors only -- ``stp3_amd.losses``
keeps the torch statement for CPU tensors and float64."""
import ctypes

import torch

from . import _lib, ops
from ._lib import check

_WS = {}


def _workspace(nbytes, device):
    key = ops._ws_key(device)                 # per device and stream (ops._ws_key)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def supported(x):
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16)


def _dtype_code(x):
    return _lib.DTYPE_BF16 if x.dtype == torch.bfloat16 else _lib.DTYPE_F32


def _dense(t):
    """Non-overlapping and dense: the strides are a permutation of a contiguous layout (no gaps)."""
    expect = 1
    for st, sz in sorted((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1):
        if st != expect:
            return False
        expect *= sz
    return True


def _rows_pixels(x):
    """(rows..., C, H, W) logits -> (tensor, rows, C, P, stride_row, stride_c, stride_p): the leading dimensions must
    collapse into one row stride and (H, W) into one pixel stride; anything else (a channel slice of a wider map) is
    copied dense first."""
    *lead, c, h, w = x.shape
    rows = 1
    for v in lead:
        rows *= v

    def collapsible(t):
        st, sz = t.stride(), t.shape
        n = len(lead)
        ok = all(st[i] == st[i + 1] * sz[i + 1] for i in range(n - 1)) if n > 1 else True
        return ok and (h == 1 or st[-2] == st[-1] * w) and _dense(t)
    if not collapsible(x):
        x = x.contiguous()
    st = x.stride()
    srow = st[len(lead) - 1] if lead else 0
    return x, rows, c, h * w, srow, st[-3], st[-1]


class _CeTopK(torch.autograd.Function):
    """out_scale * sum over rows of (sum of the k largest per-pixel weighted cross-entropies of the row)."""

    @staticmethod
    def forward(ctx, logits, labels, class_weights, row_scale, k, out_scale, ignore_index):
        ops._need_gpu(logits, labels)
        x, rows, c, p, srow, sc, sp = _rows_pixels(logits)
        dev = x.device
        labels = labels.reshape(rows, p).contiguous()
        if labels.dtype != torch.int64:
            labels = labels.long()
        w = None if class_weights is None else class_weights.to(device=dev, dtype=torch.float32).contiguous()
        rs = None if row_scale is None else row_scale.to(device=dev, dtype=torch.float32).contiguous()
        dims = _lib.CeDims(rows, p, c, int(k), int(ignore_index), _dtype_code(x), srow, sc, sp)
        lib = _lib.lib()
        need = ctypes.c_size_t()
        check(lib.stp3_ce_topk_workspace_bytes(ctypes.byref(dims), ctypes.byref(need)), 'stp3_ce_topk_workspace_bytes')
        ws = _workspace(need.value, dev)
        loss_px = torch.empty(rows, p, dtype=torch.float32, device=dev)
        sel = torch.empty(rows, 2, dtype=torch.float32, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        check(lib.stp3_ce_topk_fwd(ctypes.byref(dims), x.data_ptr(), labels.data_ptr(), ops._opt_ptr(w), ops._opt_ptr(rs),
                                   loss_px.data_ptr(), sel.data_ptr(), float(out_scale), 0, out.data_ptr(), ws.data_ptr(),
                                   need.value, ops._stream_handle()), 'stp3_ce_topk_fwd')
        ctx.save_for_backward(x, labels, w, rs, loss_px, sel)
        ctx.cfg = (dims, float(out_scale), tuple(logits.shape))
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        x, labels, w, rs, loss_px, sel = ctx.saved_tensors
        dims, out_scale, shape = ctx.cfg
        g = gout.reshape(1).to(device=x.device, dtype=torch.float32).contiguous()
        dx = torch.empty_like(x)                       # dense: same strides as x
        check(_lib.lib().stp3_ce_topk_bwd(ctypes.byref(dims), x.data_ptr(), labels.data_ptr(), ops._opt_ptr(w), ops._opt_ptr(rs),
                                          loss_px.data_ptr(), sel.data_ptr(), g.data_ptr(), out_scale, dx.data_ptr(),
                                          ops._stream_handle()), 'stp3_ce_topk_bwd')
        return dx.view(shape), None, None, None, None, None, None


def ce_topk_mean(logits, labels, class_weights=None, row_scale=None, top_k=0, ignore_index=255):
    """mean over rows and over the ``top_k`` largest pixels of each row (all pixels when ``top_k`` <= 0) of
    row_scale[row] * w[y] * cross_entropy(logits[row, :, pixel], y); logits (rows..., C, H, W), labels (rows..., H, W)."""
    *lead, c, h, w = logits.shape
    rows = 1
    for v in lead:
        rows *= v
    p = h * w
    k = int(top_k) if 0 < int(top_k) < p else 0
    return _CeTopK.apply(logits, labels, class_weights, row_scale, k, 1.0 / (rows * (k if k else p)), ignore_index)


class _RegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, row_scale, norm, ignore_value):
        ops._need_gpu(pred, target)
        *lead, c, h, w = pred.shape
        rows = 1
        for v in lead:
            rows *= v
        x = pred.contiguous()
        t = target.to(torch.float32).contiguous()
        dev = x.device
        rs = None if row_scale is None else row_scale.to(device=dev, dtype=torch.float32).contiguous()
        lib = _lib.lib()
        need = ctypes.c_size_t()
        check(lib.stp3_reg_loss_workspace_bytes(ctypes.byref(need)), 'stp3_reg_loss_workspace_bytes')
        ws = _workspace(need.value, dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        check(lib.stp3_reg_loss_fwd(rows, c, h * w, int(norm), float(ignore_value), _dtype_code(x), x.data_ptr(), t.data_ptr(),
                                    ops._opt_ptr(rs), out.data_ptr(), ws.data_ptr(), need.value, ops._stream_handle()),
              'stp3_reg_loss_fwd')
        ctx.save_for_backward(x, t, rs, out)
        ctx.cfg = (rows, c, h * w, int(norm), float(ignore_value), tuple(pred.shape))
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        x, t, rs, stat = ctx.saved_tensors
        rows, c, p, norm, ignore_value, shape = ctx.cfg
        g = gout.reshape(1).to(device=x.device, dtype=torch.float32).contiguous()
        dx = torch.empty_like(x)
        check(_lib.lib().stp3_reg_loss_bwd(rows, c, p, norm, ignore_value, _dtype_code(x), x.data_ptr(), t.data_ptr(),
                                           ops._opt_ptr(rs), stat.data_ptr(), g.data_ptr(), dx.data_ptr(),
                                           ops._stream_handle()), 'stp3_reg_loss_bwd')
        return dx.view(shape), None, None, None, None


def regression_loss(pred, target, row_scale=None, norm=1, ignore_value=255.0):
    """stp3/losses.py:6-40: mean over the pixels with target[..., 0, h, w] != ignore of row_scale * sum_c |d| or d^2;
    pred / target (rows..., C, H, W)."""
    return _RegLoss.apply(pred, target, row_scale, norm, ignore_value)


def warp_nearest(x, theta, identity=None):
    """x (F, C, H, W) float32 label maps, theta (F, 2, 3) ``F.affine_grid`` matrices (host or device), identity: sequence
    of F flags -- frames copied unchanged.  All frames and channels in one launch (stp3_warp_nearest)."""
    ops._need_gpu(x)
    f, c, h, w = x.shape
    x = x.to(torch.float32).contiguous()
    th = theta.to(device=x.device, dtype=torch.float32).reshape(f, 6).contiguous()
    ident = None
    if identity is not None:
        if torch.is_tensor(identity):
            ident = identity.to(device=x.device, dtype=torch.int32).contiguous()        # (already there: no copy)
        else:
            ident = torch.as_tensor(list(identity), dtype=torch.int32).to(x.device, non_blocking=True)
    y = torch.empty_like(x)
    check(_lib.lib().stp3_warp_nearest(f, c, h, w, x.data_ptr(), th.data_ptr(), ops._opt_ptr(ident), y.data_ptr(),
                                       ops._stream_handle()), 'stp3_warp_nearest')
    return y
