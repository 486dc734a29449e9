"""Host-side cost of the operator wrappers, measured WITHOUT a GPU.

The training step is host-bound on MI355X (GPU busy ~55 %), so the host time between two
kernel launches matters.  This script builds a stand-in for libstp3hip.so whose entry points return immediately
(size queries answer 1 MiB), points the binding at it and times forward + backward of each autograd wrapper on
tiny CPU tensors: what is left is exactly the per-call host work (argument marshalling, workspace bookkeeping,
autograd glue).  Both launch paths are measured: the Python/ctypes one and, if built, the C++ one
(csrc/host/stp3_host.cpp, STP3_CPP_OPS=1).

    python scripts/host_overhead.py
"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))


def build_mock(path):
    from stp3_amd import _lib
    src = ['#include <stddef.h>', 'const char* stp3_version(void) { return "stp3hip mock"; }']
    for name in _lib.SIGNATURES:
        if name == 'stp3_version':
            continue
        if name.endswith('_bytes') or name.endswith('_workspace'):
            src.append(f'int {name}(const void* dims, size_t* bytes) {{ *bytes = 1 << 20; return 0; }}')
        else:
            src.append(f'int {name}() {{ return 0; }}')          # K&R definition: callable with any arguments
    c = path[:-3] + '.c'
    open(c, 'w').write('\n'.join(src) + '\n')
    subprocess.check_call(['gcc', '-shared', '-fPIC', '-O1', '-w', c, '-o', path])


def timeit(fn, n=2000):
    for _ in range(50):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


def worker(mock):
    import torch
    from stp3_amd import _lib
    _lib.LIB_PATH = mock
    from stp3_amd import ops
    ops._need_gpu = lambda *a: None
    ops._stream = lambda: None
    ops._stream_handle = lambda: 0
    ops.WGRAD_MIN_CHANNELS = 0                       # every gradient through the (mock) library
    torch.set_num_threads(1)
    cl = torch.channels_last
    x = torch.randn(2, 64, 8, 8).to(memory_format=cl).requires_grad_()
    w, b = torch.ones(64, requires_grad=True), torch.zeros(64, requires_grad=True)
    rm, rv = torch.zeros(64), torch.ones(64)
    g = torch.randn(2, 64, 8, 8).to(memory_format=cl)
    xb = x.detach().bfloat16().requires_grad_()
    gb = g.bfloat16()
    wc = torch.randn(64, 64, 3, 3, requires_grad=True)
    wd = torch.randn(64, 1, 3, 3, requires_grad=True)
    res = torch.randn(2, 64, 8, 8).to(memory_format=cl).bfloat16().requires_grad_()

    def bn():
        ops.bn_act(xb, w, b, rm, rv, True, 0.1, 1e-5, act=ops.ACT_RELU, group=False).backward(gb)

    def bn_res():
        ops.bn_act(xb, w, b, rm, rv, True, 0.1, 1e-5, act=ops.ACT_NONE, res=res, res_mode=ops.RES_AFTER_ACT,
                   group=False).backward(gb)

    def conv():
        ops.conv2d(xb, wc, None, 1, 1, 1).backward(gb)

    def dw():
        ops.depthwise_conv2d(xb, wd, 1, (1, 1, 1, 1)).backward(gb)

    def torch_op():                                   # yardstick: one cheap differentiable torch op pair
        (xb * 2.0).backward(gb)

    path = 'C++ (STP3_CPP_OPS=1)' if ops._CPP is not None else 'Python/ctypes'
    print(f'launch path: {path}')
    for name, fn in [('bn_act + relu            fwd+bwd', bn), ('bn_act + residual        fwd+bwd', bn_res),
                     ('conv2d 3x3 (dx, dw)      fwd+bwd', conv), ('depthwise_conv2d        fwd+bwd', dw),
                     ('yardstick: torch mul     fwd+bwd', torch_op)]:
        print(f'  {name:36s} {timeit(fn):8.1f} us/call')


def main():
    if len(sys.argv) > 2 and sys.argv[1] == '--worker':
        worker(sys.argv[2])
        return
    with tempfile.TemporaryDirectory() as tmp:
        mock = os.path.join(tmp, 'libstp3hip_mock.so')
        build_mock(mock)
        for cpp in ('0', '1'):
            if cpp == '1' and not os.path.exists(os.path.join(ROOT, 'st-p3_amd', 'stp3_amd', '_stp3_host.so')):
                print('C++ launch path not built (python st-p3_amd/csrc/host/build_host.py)')
                continue
            env = dict(os.environ, STP3_CPP_OPS=cpp, STP3_HOST_DRYRUN='1')
            subprocess.check_call([sys.executable, os.path.abspath(__file__), '--worker', mock], env=env)


if __name__ == '__main__':
    main()
