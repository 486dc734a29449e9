"""Live timing of the C-ABI calls with events recorded on the stream each call is launched on, and the ALGORITHMIC work
of every call (flops of the convolutions, compulsory bytes of the streaming kernels) -- the per-family rooflines of
``bench.py``'s JSON line (MFMA utilisation of the convolutions = 2 * MAC / time / 2.5 PF; HBM GB/s of
the bandwidth-bound families against 8 TB/s).

Off by default: ``_lib.lib()`` hands out the raw ctypes handle and nothing here runs.  ``enable()`` makes it hand out a
proxy that brackets every call listed in ``WORK`` with two events; ``summary()`` synchronises and aggregates per family.
"""
import torch

ENABLED = False
RECORDS = []          # (family, work, start_event, end_event, shape label or None, compulsory bytes or None)


def _obj(arg):
    """The structure behind a ctypes.byref(...) argument."""
    return getattr(arg, '_obj', arg)


def _esize(dtype_code):
    return 2 if dtype_code == 1 else 4


def _conv_flops(args):
    d = _obj(args[0])
    return 2.0 * d.N * d.Ho * d.Wo * d.Cout * d.KH * d.KW * d.Cin


def _conv_bytes(args):
    """Compulsory bytes of one convolution call: input, output (bf16 / float32) and weights once."""
    d = _obj(args[0])
    return (2.0 * d.N * d.H * d.W * d.Cin + float(_esize(1 if d.out_dtype == 1 else 0)) * d.N * d.Ho * d.Wo * d.Cout
            + 2.0 * d.Cout * d.KH * d.KW * d.Cin)


def _wgrad_bytes(args):
    """... of one weight-gradient call: both bf16 operands and the float32 result once."""
    d = _obj(args[0])
    return 2.0 * d.N * d.H * d.W * d.Cin + 2.0 * d.N * d.Ho * d.Wo * d.Cout + 4.0 * d.Cout * d.KH * d.KW * d.Cin


def _conv_label(args):
    d = _obj(args[0])
    return (f'{d.KH}x{d.KW}' + (f'/{d.stride}' if d.stride != 1 else '') + (f' d{d.dil_h}' if d.dil_h != 1 else '') +
            f' {d.Cin}->{d.Cout} @{d.Ho}x{d.Wo} x{d.N}')


LABEL = {'stp3_conv2d_fwd': _conv_label, 'stp3_conv2d_fwd_add': _conv_label, 'stp3_conv2d_wgrad': _conv_label, 'stp3_conv2d_wgrad_partials': _conv_label}


def _bn_bytes(tensors):
    def f(args):
        d = _obj(args[0])
        n = tensors + (1 if d.res_mode else 0)
        return float(n) * d.N * d.rows * d.C * _esize(d.dtype)
    return f


def _dw_bytes(args):
    d = _obj(args[0])
    return float(d.N * d.C * (d.H * d.W + d.Ho * d.Wo)) * _esize(d.dtype)


def _se_bytes(tensors):
    def f(args):
        d = _obj(args[0])
        return float(tensors) * d.N * d.rows * d.C * _esize(d.dtype)
    return f


def _se_pool_bytes(args):
    d = _obj(args[0])
    return float(2 if args[2] else 1) * d.N * d.rows * d.C * _esize(d.dtype)


# C-ABI entry -> (family, algorithmic work of one call); flops for 'conv*', bytes for everything else
WORK = {
    'stp3_conv2d_fwd': ('conv_fwd_dgrad', _conv_flops),
    'stp3_conv2d_fwd_add': ('conv_fwd_dgrad', _conv_flops),                 # (a data gradient + the skip's gradient)
    # (the expand convolution's data gradient inside the BatchNorm-backward apply pass of the recomputing route: counted with
    # the data gradient's flops and the WHOLE pass's time -- pessimistic for the family, never flattering)
    'stp3_conv2d_bn_bwd_apply_dx': ('conv_fwd_dgrad', _conv_flops),
    'stp3_conv2d_wgrad': ('conv_wgrad', _conv_flops),
    'stp3_conv2d_wgrad_partials': ('conv_wgrad', _conv_flops),              # (the split contraction: all of the layer's flops)
    'stp3_conv2d_wgrad_reduce_batch': ('conv_wgrad', lambda args: 0.0),     # (their deferred sums, one launch per pass: time only)
    'stp3_bn_stats': ('batchnorm', _bn_bytes(1)),
    'stp3_bn_apply_fwd': ('batchnorm', _bn_bytes(2)),
    'stp3_bn_fwd_train': ('batchnorm', _bn_bytes(2)),          # statistics + apply over the same x: x is compulsory once
    'stp3_bn_bwd_reduce': ('batchnorm', _bn_bytes(0)),         # the apply pass below needs dy, x anyway: no new bytes
    'stp3_bn_apply_bwd': ('batchnorm', _bn_bytes(3)),
    'stp3_bn_bwd_train': ('batchnorm', _bn_bytes(3)),
    'stp3_dwconv2d_fwd': ('depthwise', _dw_bytes),
    'stp3_dwconv2d_fwd_stats': ('depthwise', _dw_bytes),
    'stp3_dwconv2d_fwd_stats_bn': ('depthwise', _dw_bytes),
    'stp3_dwconv2d_bwd_data': ('depthwise', _dw_bytes),
    'stp3_dwconv2d_bwd_weight': ('depthwise', _dw_bytes),
    'stp3_dwconv2d_bwd_weight_oihw': ('depthwise', _dw_bytes),
    'stp3_se_pool': ('squeeze_excite', _se_pool_bytes),
    'stp3_se_scale': ('squeeze_excite', _se_bytes(2)),
    'stp3_se_pool_act': ('mbconv', _se_bytes(1)),
    'stp3_mbconv_scale_act': ('mbconv', _se_bytes(2)),
    'stp3_mbconv_bwd_reduce': ('mbconv', _se_bytes(2)),
    'stp3_mbconv_bwd_apply': ('mbconv', _se_bytes(3)),
}


class TimedLib:
    def __init__(self, handle):
        self._handle = handle

    def __getattr__(self, name):
        fn = getattr(self._handle, name)
        spec = WORK.get(name)
        if spec is None:
            wrapped = fn
        else:
            family, work = spec
            label = LABEL.get(name)

            def wrapped(*args):
                start = torch.cuda.Event(enable_timing=True)
                end = torch.cuda.Event(enable_timing=True)
                start.record()
                rc = fn(*args)
                end.record()
                RECORDS.append((family, work(args), start, end, label(args) if label else None,
                                (_wgrad_bytes if 'wgrad' in name else _conv_bytes)(args) if label else None))
                return rc
        self.__dict__[name] = wrapped
        return wrapped


def enable(on=True):
    global ENABLED
    ENABLED = bool(on)
    if on:
        RECORDS.clear()


def summary():
    """family -> {'calls', 'ms', 'work'} over everything recorded since ``enable()`` (synchronises)."""
    torch.cuda.synchronize()
    out = {}
    for family, work, s, e, _, _ in RECORDS:
        a = out.setdefault(family, {'calls': 0, 'ms': 0.0, 'work': 0.0})
        a['calls'] += 1
        a['ms'] += s.elapsed_time(e)
        a['work'] += work
    return out


def by_shape(top=12):
    """The labelled calls (the convolutions) aggregated per (family, shape), the `top` most expensive first:
    [{'family', 'shape', 'calls', 'ms', 'work'}] over everything recorded since ``enable()`` (synchronises)."""
    torch.cuda.synchronize()
    out = {}
    for family, work, s, e, label, nbytes in RECORDS:
        if label is None:
            continue
        a = out.setdefault((family, label), {'family': family, 'shape': label, 'calls': 0, 'ms': 0.0, 'work': 0.0, 'bytes': 0.0})
        a['calls'] += 1
        a['ms'] += s.elapsed_time(e)
        a['work'] += work
        a['bytes'] += nbytes or 0.0
    return sorted(out.values(), key=lambda a: -a['ms'])[:top]
