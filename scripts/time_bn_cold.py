"""Do the streaming BatchNorm kernels run slower inside the step than alone because their operands are COLD there?
Each kernel is timed over a rotation of R buffer sets (R x bytes >> the 256 MB memory-side cache: every launch streams from
HBM, like in the step) and over ONE set (warm: what scripts/time_bn.py measures), beside torch's copy of the same bytes."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import torch
from stp3_amd import _lib
if os.environ.get('EXP_LIB'):
    _lib.LIB_PATH = os.environ['EXP_LIB']
from stp3_amd import ops

SHAPES = [('bev 64ch', 12, 64, 200, 200), ('bev 128ch', 12, 128, 200, 200), ('trunk b1 project 24ch', 72, 24, 112, 240),
          ('trunk b3 project 32ch', 72, 32, 56, 120), ('trunk b3 expanded 192ch', 72, 192, 56, 120)]
R = int(os.environ.get('ROTATE', '10'))


def ev(fn, iters):
    fn(0); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


lib = _lib.lib()
stream = ops._stream_handle()
for name, n, c, h, w in SHAPES:
    mk = lambda: torch.randn(n, h, w, c, device='cuda').to(torch.bfloat16)
    xs, ys, gs = [mk() for _ in range(R)], [mk() for _ in range(R)], [mk() for _ in range(R)]
    dims = _lib.BnDims(n, h * w, c, c, c, c, _lib.DTYPE_BF16, ops.ACT_RELU, ops.RES_NONE, 0, 0, 0)
    ws, ws_bytes = ops._bn_workspace(n, c, xs[0].device)
    stat = torch.zeros(4 * c, device='cuda'); gam = torch.ones(c, device='cuda'); bet = torch.zeros(c, device='cuda')
    rm = torch.zeros(c, device='cuda'); rv = torch.ones(c, device='cuda')
    sums = torch.zeros(3 * c, device='cuda'); ssum = torch.zeros(n * 3 * c, device='cuda')
    base = stat.data_ptr()
    _lib.check(lib.stp3_bn_stats(ctypes.byref(dims), xs[0].data_ptr(), None, ws.data_ptr(), ws_bytes, base, stream), 'stats')
    cnt = float(n * h * w)

    def apply_fwd(i, rot=True):
        k = i % R if rot else 0
        lib.stp3_bn_apply_fwd(ctypes.byref(dims), xs[k].data_ptr(), None, None, None, base, cnt, gam.data_ptr(), bet.data_ptr(), 1e-5,
                              0.1, rm.data_ptr(), rv.data_ptr(), base + 8 * c, base + 12 * c, ys[k].data_ptr(), stream)

    def bwd_reduce(i, rot=True):
        k = i % R if rot else 0
        lib.stp3_bn_bwd_reduce(ctypes.byref(dims), gs[k].data_ptr(), xs[k].data_ptr(), None, None, None, base + 8 * c, base + 12 * c,
                               gam.data_ptr(), bet.data_ptr(), ws.data_ptr(), ws_bytes, ssum.data_ptr(), sums.data_ptr(), stream)

    def apply_bwd(i, rot=True):
        k = i % R if rot else 0
        lib.stp3_bn_apply_bwd(ctypes.byref(dims), gs[k].data_ptr(), xs[k].data_ptr(), None, None, None, base + 8 * c, base + 12 * c,
                              gam.data_ptr(), bet.data_ptr(), sums.data_ptr(), cnt, ys[k].data_ptr(), None, stream)

    def stats(i, rot=True):
        k = i % R if rot else 0
        lib.stp3_bn_stats(ctypes.byref(dims), xs[k].data_ptr(), None, ws.data_ptr(), ws_bytes, base, stream)

    def tcopy(i, rot=True):
        k = i % R if rot else 0
        ys[k].copy_(xs[k])

    def tadd(i, rot=True):
        k = i % R if rot else 0
        torch.add(xs[k], gs[k], out=ys[k])

    nb = n * c * h * w * 2
    print(f'{name}: tensor {nb/1e6:.1f} MB, rotation of {R}')
    for label, fn, passes in (('torch copy (R+W)', tcopy, 2), ('torch add (2R+W)', tadd, 3), ('bn_stats + reduce (R)', stats, 1),
                              ('bn_apply_fwd (R+W)', apply_fwd, 2), ('bn_bwd_reduce + reduce (2R)', bwd_reduce, 2),
                              ('bn_apply_bwd (2R+W)', apply_bwd, 3)):
        cold = ev(lambda i: fn(i, True), 4 * R)
        warm = ev(lambda i: fn(i, False), 4 * R)
        print(f'   {label:30s} cold {cold:7.1f} us {passes*nb/cold/1e6:6.2f} TB/s | warm {warm:7.1f} us {passes*nb/warm/1e6:6.2f} TB/s')
    del xs, ys, gs
