"""CPU test infrastructure: record what the host code hands to the C ABI.

``build_recorder(path)`` compiles (gcc) a stand-in for libstp3hip.so generated from the prototypes of
include/stp3_hip.h.  Every entry point appends one line to $STP3_TRACE_LOG -- name, the dims struct as hex, every
scalar argument, and per pointer argument ``N`` (null) / its alias class within the call, plus a checksum of the
first bytes of the INPUT buffers that hold caller data (activations, gradients, weights, BatchNorm parameters) --
and returns 0 without touching a GPU; the size queries are answered by the real library ($STP3_REAL_LIB).

Run as a script it is the *driver*: it pushes a fixed set of operator calls (forward + backward) through
``stp3_amd.ops`` on small CPU tensors.  tests/test_host_paths_cpu.py runs the driver once per launch path
(Python/ctypes, C++ extension) and compares the two traces: identical traces mean the C++ path drives the
(GPU-validated) kernels exactly like the Python path does.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'stp3_hip.h')

# input buffers whose contents come from the caller (not from an earlier, here mocked, kernel)
DATA_PARAMS = {'x', 'dy', 'w', 'res', 'gamma', 'beta', 'bias', 'sbias', 'oscale', 'grad_out', 'seg_off', 'gate', 'add'}


def prototypes():
    src = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    out = []
    for m in re.finditer(r'\b(int|const char\*)\s+(stp3_\w+)\s*\(([^)]*)\)\s*;', src):
        params = [' '.join(p.split()) for p in m.group(3).split(',')]
        out.append((m.group(1), m.group(2), [] if params == ['void'] else params))
    return out


def recorder_source():
    lines = ['#include <dlfcn.h>', '#include <stdint.h>', '#include <stdio.h>', '#include <stdlib.h>',
             '#include <string.h>', '#include "stp3_hip.h"', '',
             'static FILE* logf(void) { static FILE* f; if (!f) f = fopen(getenv("STP3_TRACE_LOG"), "a"); return f; }',
             'static void* real(const char* name) { static void* h; if (!h) h = dlopen(getenv("STP3_REAL_LIB"), RTLD_NOW);'
             ' return dlsym(h, name); }',
             'static void hex(FILE* f, const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p;'
             ' for (size_t i = 0; i < n; ++i) fprintf(f, "%02x", b[i]); }',
             'static unsigned sum32(const void* p) { const unsigned char* b = (const unsigned char*)p; unsigned s = 0;'
             ' for (int i = 0; i < 32; ++i) s = s * 131u + b[i]; return s; }', '']
    for ret, name, params in prototypes():
        if name == 'stp3_version':
            lines.append('const char* stp3_version(void) { return "stp3hip recorder"; }')
            continue
        names = [p.split('*')[-1].split()[-1] for p in params]
        lines.append(f'{ret} {name}({", ".join(params)}) {{')
        lines.append(f'    FILE* f = logf(); fprintf(f, "{name}");')
        if name.endswith('_bytes') or name.endswith('_workspace'):      # size queries: answered by the real library
            types = ', '.join(p.rsplit(' ', 1)[0] if '*' not in p.rsplit(' ', 1)[1] else p.rsplit('*', 1)[0] + '*'
                              for p in params)
            lines.append(f'    int rc = ((int (*)({types}))real("{name}"))({", ".join(names)});')
            for p, n in zip(params[:-1], names[:-1]):
                if '_dims*' in p:
                    lines.append(f'    fprintf(f, " {n}="); hex(f, {n}, sizeof(*{n}));')
                else:
                    lines.append(f'    fprintf(f, " {n}=%lld", (long long){n});')
            lines.append(f'    fprintf(f, " -> rc=%d bytes=%zu\\n", rc, *{names[-1]}); fflush(f); return rc;')
            lines.append('}')
            continue
        ptrs = [n for p, n in zip(params, names) if '*' in p and '_dims*' not in p and n != 'stream']
        for p, n in zip(params, names):
            if '_dims*' in p:
                lines.append(f'    fprintf(f, " {n}="); hex(f, {n}, sizeof(*{n}));')
            elif n == 'stream':
                continue
            elif '*' in p:
                k = ptrs.index(n)
                alias = ' '.join(f'else if ((const void*){n} == (const void*){ptrs[j]}) fprintf(f, " {n}=A{j}");'
                                 for j in range(k))
                lines.append(f'    if (!{n}) fprintf(f, " {n}=N"); {alias} else fprintf(f, " {n}=A{k}");')
                if n in DATA_PARAMS:
                    lines.append(f'    if ({n}) fprintf(f, ":%08x", sum32({n}));')
            elif p.startswith('float') or p.startswith('double'):
                lines.append(f'    fprintf(f, " {n}=%.9g", (double){n});')
            else:
                lines.append(f'    fprintf(f, " {n}=%lld", (long long){n});')
        lines.append('    fprintf(f, "\\n"); fflush(f); return 0;')
        lines.append('}')
    return '\n'.join(lines) + '\n'


def build_recorder(path):
    c = path[:-3] + '.c'
    with open(c, 'w') as f:
        f.write(recorder_source())
    subprocess.check_call(['gcc', '-shared', '-fPIC', '-O1', '-w', '-I', os.path.join(ROOT, 'include'), c, '-o', path,
                           '-ldl'])
    return path


# ------------------------------------------------------------------------------------------------------------------
# driver
# ------------------------------------------------------------------------------------------------------------------
def drive(recorder):
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
    from stp3_amd import _lib
    _lib.LIB_PATH = recorder
    from stp3_amd import ops
    ops._need_gpu = lambda *a: None
    ops._stream = lambda: None
    ops._stream_handle = lambda: 0
    torch.set_num_threads(1)
    cl = torch.channels_last
    log = open(os.environ['STP3_TRACE_LOG'], 'a')

    def mark(text):
        log.write(f'# {text}\n')
        log.flush()

    def grads(**tensors):
        # what autograd handed back: presence, shape and dtype per input (the values come from mocked kernels)
        parts = []
        for k, t in tensors.items():
            if t is None:
                continue
            g = t.grad
            parts.append(f'{k}:' + ('none' if g is None else f'{tuple(g.shape)}/{g.dtype}/{g.stride()}'))
        mark('grads ' + ' '.join(parts))

    def rnd(*shape, dtype=torch.float32, grad=True, seed=[0]):
        seed[0] += 1
        g = torch.Generator().manual_seed(seed[0])
        t = torch.randn(*shape, generator=g)
        if t.dim() == 4:
            t = t.to(memory_format=cl)
        t = t.to(dtype)
        return t.requires_grad_() if grad else t

    # fused BatchNorm + activation
    n, c, h, w = 8, 24, 6, 10          # n >= 8: the recorder checksums the first 32 bytes of every data buffer
    for dtype in (torch.float32, torch.bfloat16):
        for training in (True, False):
            for case, kw in [('relu', dict(act=ops.ACT_RELU)),
                             ('swish+res_before', dict(act=ops.ACT_SWISH, res_mode=ops.RES_BEFORE_ACT)),
                             ('none+res_after+oscale', dict(act=ops.ACT_NONE, res_mode=ops.RES_AFTER_ACT, oscale=True)),
                             ('relu+sbias', dict(act=ops.ACT_RELU, sbias=True)),
                             ('none, no affine, no running stats', dict(act=ops.ACT_NONE, affine=False))]:
                mark(f'bn_act {dtype} training={training} {case}')
                x = rnd(n, c, h, w, dtype=dtype)
                affine = kw.get('affine', True)
                weight = rnd(c) if affine else None
                bias = rnd(c) if affine else None
                rm = torch.zeros(c) if affine else None
                rv = torch.ones(c) if affine else None
                if not affine and not training:
                    continue
                res = rnd(n, c, h, w, dtype=dtype) if kw.get('res_mode') else None
                sbias = rnd(n, c) if kw.get('sbias') else None
                oscale = rnd(n, grad=False).abs() if kw.get('oscale') else None
                y = ops.bn_act(x, weight, bias, rm, rv, training, 0.1, 1e-3, act=kw['act'], res=res,
                               res_mode=kw.get('res_mode', ops.RES_NONE), sbias=sbias, oscale=oscale, group=False)
                y.backward(rnd(n, c, h, w, dtype=dtype, grad=False))
                mark(f'out {tuple(y.shape)}/{y.dtype}/{y.stride()}')
                grads(x=x, weight=weight, bias=bias, res=res, sbias=sbias)
    # a non-channels-last input and a channel slice (leading dimension > C)
    mark('bn_act NCHW-contiguous input')
    x = torch.randn(2, 16, 5, 7, generator=torch.Generator().manual_seed(99)).requires_grad_()
    ops.bn_act(x, rnd(16), rnd(16), torch.zeros(16), torch.ones(16), True, 0.1, 1e-5, act=ops.ACT_RELU,
               group=False).backward(rnd(2, 16, 5, 7, grad=False))
    grads(x=x)
    mark('bn_act channel slice of a wider tensor')
    wide = rnd(2, 40, 5, 7, grad=False)
    xs = wide[:, 8:24].detach().requires_grad_()
    ops.bn_act(xs, rnd(16), rnd(16), torch.zeros(16), torch.ones(16), True, 0.1, 1e-5, act=ops.ACT_RELU,
               group=False).backward(rnd(2, 16, 5, 7, grad=False))
    grads(x=xs)

    # dense convolution (forward, data gradient, weight gradient)
    for wg_min in (0, 128):
        ops.WGRAD_MIN_CHANNELS = wg_min
        for case, (cin, cout, k, s, p, d, bias) in {
                '3x3': (64, 64, 3, 1, 1, 1, False), '1x1 bias': (32, 144, 1, 1, 0, 1, True),
                '3x3 stride 2': (64, 128, 3, 2, 1, 1, False), '3x3 dilation 12': (160, 160, 3, 1, 12, 12, False),
                '7x7 stride 2': (64, 64, 7, 2, 3, 1, False), '3x3 odd channels': (35, 35, 3, 1, 1, 1, True)}.items():
            x = rnd(2, cin, 12, 20, dtype=torch.bfloat16)
            wt = rnd(cout, cin, k, k)
            if wt.shape[1] % 8 != 0:                  # ops.conv2d_supported minus its is_cuda test
                mark(f'conv2d {case}: unsupported, skipped')
                continue
            mark(f'conv2d {case} wgrad_min={wg_min}')
            b = rnd(cout) if bias else None
            y = ops.conv2d(x, wt, b, s, p, d)
            y.backward(rnd(*y.shape, dtype=torch.bfloat16, grad=False))
            mark(f'out {tuple(y.shape)}/{y.dtype}/{y.stride()}')
            grads(x=x, weight=wt, bias=b)
    mark('conv2d float32 output')
    x = rnd(2, 64, 12, 20, dtype=torch.bfloat16)
    y = ops.conv2d(x, rnd(8, 64, 1, 1), rnd(8), 1, 0, 1, out_dtype=torch.float32)
    y.backward(rnd(*y.shape, grad=False))

    # depthwise convolution ("static same" padding can be asymmetric)
    for dtype in (torch.bfloat16, torch.float32):
        for case, (k, s, pad) in {'k3 s1': (3, 1, (1, 1, 1, 1)), 'k5 s2 asym': (5, 2, (1, 2, 1, 2)),
                                  'k3 s2 asym': (3, 2, (0, 1, 0, 1))}.items():
            mark(f'depthwise {dtype} {case}')
            x = rnd(2, 48, 12, 20, dtype=dtype)
            y = ops.depthwise_conv2d(x, rnd(48, 1, k, k), s, pad)
            y.backward(rnd(*y.shape, dtype=dtype, grad=False))
            mark(f'out {tuple(y.shape)}/{y.dtype}/{y.stride()}')
            grads(x=x)
    mark('end')


if __name__ == '__main__':
    drive(sys.argv[1])
