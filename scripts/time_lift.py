"""HIP-event timing of the voxel-pool kernels through the C ABI.
    python scripts/time_lift.py [B=4]                 BASELINE configs[1]/[2] shape: 224x480, D=48, 200x200 BEV, T=3
    python scripts/time_lift.py <B> stress [T=5]      BASELINE configs[4] per-GPU shard: 896x1600, D=64, 400x400 BEV"""
import sys, os, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import torch
from stp3_amd import _lib
if os.environ.get('EXP_LIB'):          # experiment builds of the library (scripts/gpu_exp.sh)
    _lib.LIB_PATH = os.environ['EXP_LIB']
from stp3_amd import ops
from stp3_amd import synthetic


def ev_time(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STRESS = len(sys.argv) > 2 and sys.argv[2] == 'stress'
T = int(sys.argv[3]) if len(sys.argv) > 3 else (5 if STRESS else 3)
kw = dict(final_dim=(896, 1600), d_bound=(2.0, 66.0, 1.0), x_bound=(-50.0, 50.0, 0.25), y_bound=(-50.0, 50.0, 0.25)) if STRESS else {}
print(f'config: {"c5 stress 896x1600 D=64 BEV 400x400" if STRESS else "c2/c3 224x480 D=48 BEV 200x200"}  B={B} T={T}')
(frustum, res, start, dim), intr, extr, ego, feat, logits = synthetic.lift_case(batch=B, seq=T, seed=31, **kw)
grid = ops.LiftGrid(frustum, res, start, dim, 'cuda')
t0 = time.time(); plan = ops.LiftPlan.build(grid, intr, extr, ego, 64); torch.cuda.synchronize(); print('first plan build (host+dev) s', time.time() - t0)
d = plan.dims
lib = _lib.lib()
mats = [m.cuda() for m in ops.lift_matrices(intr, extr, ego)]
print('voxel_index alone us', ev_time(lambda: ops.voxel_index(grid, d, *mats, order=1)))
print('plan build (index + count, scan, fill, order; device part) us', ev_time(lambda: ops.LiftPlan.build(grid, intr, extr, ego, 64, out=plan)))
f = feat.cuda().permute(0, 1, 2, 4, 5, 3).reshape(d.BT, d.NPIX, d.C).contiguous()
l = logits.cuda().permute(0, 1, 2, 4, 5, 3).reshape(d.BT, d.NPIX, d.D).contiguous()
prob = torch.empty(d.BT, d.N * d.fW, d.D, d.fH, device='cuda')
needs = ctypes.c_int()
lib.stp3_lift_bwd_needs_prob(ctypes.byref(d), ctypes.byref(needs))
prob_ptr = ops._ptr(prob) if needs.value else None          # the matrix-core backward recomputes the probabilities
print('stand-alone softmax operator us', ev_time(lambda: ops.depth_softmax(d, l)))
print('runs per (b,t):', plan.offsets()[:, -1].tolist(), ' points per (b,t):', d.P, ' total runs:', int(plan.column_offsets()[-1]))
alg = d.BT * (d.NPIX * d.C * 4 + d.NPIX * d.D * 4 + d.C * d.V * 4)
algb = d.BT * (d.C * d.V * 4 + 2 * d.NPIX * d.C * 4 + 2 * d.NPIX * d.D * 4)
gf = torch.empty_like(f); gl = torch.empty_like(l)
for name, layout in (('channels-last bf16 BEV (bench path under autocast)', _lib.BEV_CHANNELS_LAST_BF16),
                     ('channels-last float32 BEV', ops.BEV_CHANNELS_LAST), ('reference layout (+ transpose pass)', ops.BEV_CHANNELS_FIRST)):
    bev = torch.empty(d.B * d.T * d.C * d.X * d.Y, device='cuda')
    ws, wsb = ops.lift_workspace(d, 'cuda')
    fwd = lambda: lib.stp3_lift_splat_fwd(ctypes.byref(d), ops._ptr(f), ops._ptr(l), ops._ptr(plan.plan), ctypes.c_float(0.5), layout, ops._ptr(ws), ctypes.c_size_t(wsb), prob_ptr, ops._ptr(bev), ops._stream())
    us = ev_time(fwd)
    print(f'{name}: lift_splat_fwd (logits -> BEV) us {us:.1f}  algorithmic {alg/1e6:.1f} MB -> {alg/us/1e6:.3f} TB/s ({alg/us/1e6/8*100:.1f}% of 8 TB/s)')
    gb = torch.randn_like(bev)
    blayout, gdt = layout, 0
    if layout == _lib.BEV_CHANNELS_LAST_BF16:          # the gradient of a bf16 BEV arrives in bf16, channels-last
        gb, blayout, gdt = gb.to(torch.bfloat16), ops.BEV_CHANNELS_LAST, 1
    bwd = lambda: lib.stp3_lift_splat_bwd(ctypes.byref(d), ops._ptr(gb), blayout, gdt, ops._ptr(f), ops._ptr(l), prob_ptr, ops._ptr(plan.vox_cm), ops._ptr(plan.plan), ctypes.c_float(0.5), ops._ptr(ws), ctypes.c_size_t(wsb), ops._ptr(gf), ops._ptr(gl), ops._stream())
    us = ev_time(bwd)
    print(f'{name}: lift_splat_bwd us {us:.1f}  algorithmic {algb/1e6:.1f} MB -> {algb/us/1e6:.3f} TB/s ({algb/us/1e6/8*100:.1f}% of 8 TB/s)')
