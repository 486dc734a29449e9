"""TEST INFRASTRUCTURE -- run one operator of stp3_amd.ops on CPU tensors through libstp3hip_cpu.so (the real kernel
sources executed on CPU threads, tests/hipcpu/build.py) and print the deviations from the oracle / torch as JSON.

    python tests/hipcpu/run_case.py <libstp3hip_cpu.so> <case>

Driver of tests/test_kernels_on_cpu.py; one process per case (the fiber order of the stand-in, HIPCPU_ORDER, is read
from the environment once)."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def setup(lib_path):
    from stp3_amd import _lib
    _lib.LIB_PATH = lib_path
    from stp3_amd import ops
    ops._need_gpu = lambda *a: None
    ops._stream = lambda: None
    ops._stream_handle = lambda: 0
    torch.Tensor.is_cuda = property(lambda self: True)            # operators take their GPU route
    return ops


def err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max())


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# ------------------------------------------------------------------------------------------------------------------
def case_lift(ops, cfg, batch, seq, cams, seed, golden=None, roll=0.0):
    from oracle import lift_oracle as lo
    from tests import helpers as H
    if golden is not None:
        g = H.load(golden)
        intr, extr, ego = (torch.from_numpy(g[k]) for k in ('intrinsics', 'extrinsics', 'future_egomotion'))
        feat, logits = torch.from_numpy(g['feat']), torch.from_numpy(g['depth_logits'])
    else:
        intr, extr, ego, feat, logits = H.lift_inputs(cfg, batch, seq, cams, seed=seed)
    if roll:                            # cameras rolled about their optical axis: the rows of an image column fan out over the grid
        c, s_ = float(np.cos(roll)), float(np.sin(roll))
        extr = extr.clone()
        extr[..., :3, :3] = extr[..., :3, :3] @ torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
    frustum, res, start, dim = H.grid_params(cfg)
    grid = ops.LiftGrid(frustum, res, start, dim, 'cpu')
    b, s, n = intr.shape[:3]
    dims = ops.make_dims(b, s, n, grid.D, grid.fH, grid.fW, cfg['out_channels'], grid.X, grid.Y, grid.Z)
    ids = ops.voxel_index(grid, dims, *ops.lift_matrices(intr, extr, ego), order=ops.VOX_REFERENCE)
    ids = ids.view(b, s, n, grid.D, grid.fH, grid.fW).numpy()
    vox = H.oracle_vox(cfg, intr, extr, ego)
    plan = ops.LiftPlan.build(grid, intr, extr, ego, cfg['out_channels'])
    pm = plan.voxel_ids().numpy()
    f, lg = feat.clone().requires_grad_(), logits.clone().requires_grad_()
    bev = ops.lift_splat(f, lg, plan, cfg['discount'])
    again = ops.lift_splat(feat, logits, plan, cfg['discount'])
    go = torch.randn(bev.shape, generator=torch.Generator().manual_seed(seed + 5))
    bev.backward(go)
    # the channels-last layout (model path: no transpose passes) gives the same bits
    f2, lg2 = feat.clone().requires_grad_(), logits.clone().requires_grad_()
    bev_cl = ops.lift_splat(f2, lg2, plan, cfg['discount'], True)
    bev_cl.backward(go)
    cl_equal = bool(torch.equal(bev_cl.detach(), bev.detach()) and torch.equal(f2.grad, f.grad) and torch.equal(lg2.grad, lg.grad)
                    and bev_cl.permute(0, 1, 3, 4, 2).is_contiguous())
    # a bfloat16 gradient in channels-last memory (what a bf16 consumer hands back) is imported directly
    go16 = go.permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16).permute(0, 1, 4, 2, 3)
    f3, lg3 = feat.clone().requires_grad_(), logits.clone().requires_grad_()
    ops.lift_splat(f3, lg3, plan, cfg['discount'], True).backward(go16)
    f4, lg4 = feat.clone().requires_grad_(), logits.clone().requires_grad_()
    ops.lift_splat(f4, lg4, plan, cfg['discount'], True).backward(go16.float())
    cl_equal = cl_equal and bool(torch.equal(f3.grad, f4.grad) and torch.equal(lg3.grad, lg4.grad))
    # bf16 BEV output (STP3_BEV_CHANNELS_LAST_BF16): the float32 result rounded once, and a backward that does not care
    f5, lg5 = feat.clone().requires_grad_(), logits.clone().requires_grad_()
    bev16 = ops.lift_splat(f5, lg5, plan, cfg['discount'], True, torch.bfloat16)
    bev16.backward(go16)
    cl_equal = cl_equal and bool(bev16.dtype == torch.bfloat16 and torch.equal(bev16.detach(), bev_cl.detach().to(torch.bfloat16))
                                 and torch.equal(f5.grad, f3.grad) and torch.equal(lg5.grad, lg3.grad))
    exact = lo.pool_exact(feat, logits, vox, (grid.X, grid.Y), cfg['discount'])
    gf, gl = lo.pool_backward_exact(go, feat, logits, vox, cfg['discount'])
    out = {'ids_equal_oracle': bool(np.array_equal(ids, vox)), 'pixel_major_ids_equal': bool(np.array_equal(pm, vox)),
           'valid_fraction': float((vox >= 0).mean()), 'fwd_err': err(bev.detach(), exact), 'fwd_scale': float(exact.abs().max()),
           'reproducible': bool(torch.equal(bev.detach(), again)), 'dfeat_err': err(f.grad, gf), 'dlogit_err': err(lg.grad, gl),
           'grad_scale': float(min(gf.abs().max(), gl.abs().max())), 'dims': [grid.D, grid.fH, grid.fW, cfg['out_channels']],
           'channels_last_equal': cl_equal, 'max_runs_per_voxel': int(plan.offsets().diff(dim=1).max()),
           'counts_clean': bool(int(plan.counts.abs().max()) == 0),
           'max_runs_per_column': int(plan.column_offsets().diff().max())}
    try:
        out['plan_runs'] = H.check_plan_structure(plan, vox)
        out['plan_ok'] = True
    except AssertionError as e:
        out['plan_ok'], out['plan_error'] = False, str(e)
    if golden is not None:
        out['ids_equal_reference'] = bool(np.array_equal(ids, g['ref_vox']))
        out['fwd_err_reference'] = err(bev.detach(), g['ref_bev'])
    return out


def lift_small(ops):
    from tests import helpers as H
    return case_lift(ops, H.SMALL, 2, 3, 2, 0, golden='lift_small.npz')


def lift_c16(ops):                  # 16 channels, 8 image rows: 8 columns per wave in the backward
    from tests import helpers as H
    return case_lift(ops, dict(H.SMALL, out_channels=16, final_dim=(64, 48)), 1, 2, 2, 3)


def lift_c16_rows32(ops):           # the same with 32 image rows per column (2 columns per wave)
    from tests import helpers as H
    return case_lift(ops, dict(H.SMALL, out_channels=16, final_dim=(64, 48), downsample=2), 1, 2, 2, 3)


def lift_c64_many_runs(ops):        # 64 channels, ~190 runs per column: more runs per depth bin than the backward stages (per-lane path)
    from tests import helpers as H
    cfg = dict(H.SMALL, out_channels=64, final_dim=(64, 16), downsample=2, d_bound=(2.0, 34.0, 1.0),
               x_bound=(-36.0, 36.0, 0.25), y_bound=(-36.0, 36.0, 0.25))
    return case_lift(ops, cfg, 1, 1, 1, 7)


def lift_c64_rolled(ops):           # 64 channels, 32 rows, 64 bins, cameras rolled about their optical axis: up to ~1 800 runs per column and
    from tests import helpers as H  # dozens of runs per voxel (tile loop of pass 1 far beyond four tiles, long contiguous places in pass 2)
    cfg = dict(H.SMALL, out_channels=64, final_dim=(64, 96), downsample=2, d_bound=(2.0, 34.0, 0.5),
               x_bound=(-36.0, 36.0, 0.25), y_bound=(-36.0, 36.0, 0.25))
    return case_lift(ops, cfg, 1, 1, 1, 17, roll=0.9)


def lift_full(ops):                 # the real geometry: 6 cameras x 224x480, D = 48, C = 64, BEV 200x200, T = 3 (golden digests)
    import hashlib
    from oracle import lift_oracle as lo
    from tests import helpers as H
    g = H.load('lift_full.npz')
    cfg = H.FULL
    intr, extr, ego, feat, logits = H.lift_inputs(cfg, 1, 3, 6, seed=5)
    frustum, res, start, dim = H.grid_params(cfg)
    grid = ops.LiftGrid(frustum, res, start, dim, 'cpu')
    dims = ops.make_dims(1, 3, 6, grid.D, grid.fH, grid.fW, 64, grid.X, grid.Y, grid.Z)
    ids = ops.voxel_index(grid, dims, *ops.lift_matrices(intr, extr, ego), order=ops.VOX_REFERENCE)
    ids = ids.view(1, 3, 6, grid.D, grid.fH, grid.fW).numpy()
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(ids).tobytes()).digest(), dtype=np.uint8)
    plan = ops.LiftPlan.build(grid, intr, extr, ego, 64)
    f, lg = feat.clone().requires_grad_(), logits.clone().requires_grad_()
    bev = ops.lift_splat(f, lg, plan, 0.5)
    go = torch.randn(1, 3, 64, 200, 200, generator=torch.Generator().manual_seed(17))
    bev.backward(go)
    vox = H.oracle_vox(cfg, intr, extr, ego)
    gf, gl = lo.pool_backward_exact(go, feat, logits, vox, 0.5)
    flat = bev.detach().reshape(-1)
    return {'ids_sha256_equal_reference': bool(np.array_equal(sha, g['generic_vox_sha256'])),
            'ids_equal_oracle': bool(np.array_equal(ids, vox)),
            'bev_exact_sample_err': err(flat[::257], g['generic_bev_exact_sample']),
            'bev_reference_sample_err': err(flat[::257], g['generic_bev_sample']),
            'bev_sum_err': err(bev.detach().double().sum(dim=(-1, -2))[0], g['generic_bev_sum_tc']),
            'dfeat_err': err(f.grad, gf), 'dlogit_err': err(lg.grad, gl)}


def lift_c64_frames(ops):           # 64 channels, 16 rows, 3 frames: the matrix-core kernels with the discount recurrence in the backward
    from tests import helpers as H
    return case_lift(ops, dict(H.SMALL, out_channels=64, final_dim=(64, 48), downsample=4), 2, 3, 2, 13)


def lift_coarse_grid(ops):          # 4 x 4 voxels of 8 m: hundreds of runs per voxel (the wave-wide ordering of long lists)
    from tests import helpers as H
    cfg = dict(H.SMALL, out_channels=16, final_dim=(64, 48), x_bound=(-16.0, 16.0, 8.0), y_bound=(-16.0, 16.0, 8.0))
    return case_lift(ops, cfg, 1, 2, 2, 11)


def lift_tall(ops):                 # 112 rows x 64 bins per column (BASELINE configs[4] column shape): two 56-row slices per column in the backward
    from tests import helpers as H
    cfg = dict(H.FULL, out_channels=8, final_dim=(224, 32), downsample=2, d_bound=(2.0, 66.0, 1.0))
    return case_lift(ops, cfg, 1, 1, 1, 9)


def lift_c64_rows56(ops):           # 64 channels, 56 rows x 48 bins: taller than the matrix-core kernels take (general kernels at C = 64)
    from tests import helpers as H
    cfg = dict(H.FULL, out_channels=64, final_dim=(224, 32), downsample=4)
    return case_lift(ops, cfg, 1, 1, 2, 17)


# ------------------------------------------------------------------------------------------------------------------
def voxsum(ops):
    from oracle import lift_oracle as lo
    from tests import helpers as H
    g = H.load('voxsum.npz')
    out = {}
    for name in ('singles', 'onevoxel', 'onerow'):
        x = torch.tensor(g[f'{name}_x'], requires_grad=True)
        y, kept = ops.VoxelsSumming.apply(x, torch.tensor(g[f'{name}_geometry']), torch.tensor(g[f'{name}_ranks']))
        y.backward(torch.tensor(g[f'{name}_grad']))
        out[name] = {'sum_err': err(y.detach(), g[f'{name}_sum64']), 'geometry_equal': bool(np.array_equal(kept.numpy(), g[f'{name}_geomkept'])),
                     'grad_equal': bool(np.array_equal(x.grad.numpy(), g[f'{name}_gradx64']))}
    rng = np.random.default_rng(3)                                  # ragged, long voxels, 70 channels (two lane passes)
    lengths = np.concatenate([rng.integers(1, 9, 40), [130, 1, 67]])
    ranks = np.repeat(np.sort(rng.choice(5000, len(lengths), replace=False)), lengths).astype(np.int64)
    xs = rng.standard_normal((len(ranks), 70)).astype(np.float32)
    ref, _, seg = lo.voxels_summing(xs, np.zeros((len(ranks), 3)), ranks)
    x = torch.tensor(xs, requires_grad=True)
    y, _ = ops.VoxelsSumming.apply(x, torch.zeros(len(ranks), 3), torch.tensor(ranks))
    gy = rng.standard_normal(ref.shape).astype(np.float32)
    y.backward(torch.tensor(gy))
    out['ragged'] = {'sum_err': err(y.detach(), ref), 'grad_equal': bool(np.array_equal(x.grad.numpy(), lo.voxels_summing_backward(gy, seg).astype(np.float32)))}
    return out


def wprep(ops):
    torch.manual_seed(0)
    ws = [torch.nn.Parameter((torch.randn(*s) * 3).contiguous(memory_format=torch.channels_last if i % 2 else torch.contiguous_format))
          for i, s in enumerate([(8, 16, 3, 3), (24, 8, 1, 1), (5, 40, 7, 7), (64, 64, 3, 3), (3, 8, 5, 5)])]
    sh = ops._WeightShadows('cpu')
    for w in ws:
        sh.register(w)
    with torch.no_grad():
        for w in ws:
            w.mul_(-0.7)
    bad = 0
    for w in ws:
        ent = sh.lookup(w)                                      # notices the in-place update, refreshes
        rb = w.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        rt = rb.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
        bad += int(not torch.equal(ent['wb'], rb)) + int(not torch.equal(ent['wt'], rt))
    return {'mismatching_tensors': bad}


def optim(ops):
    from stp3_amd import parallel
    from stp3_amd.parallel import FlatAdam, GradientBuckets

    def make():
        torch.manual_seed(5)
        return torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3, padding=1), torch.nn.Flatten(), torch.nn.Linear(16 * 36, 40),
                                   torch.nn.ReLU(), torch.nn.Linear(40, 5))
    ref_m, fus_m = make(), make()
    ref_b, fus_b = GradientBuckets(ref_m, bucket_bytes=20000), GradientBuckets(fus_m, bucket_bytes=20000, gather=False)
    ref_o, fus_o = FlatAdam(ref_b, lr=1e-2, weight_decay=1e-3), FlatAdam(fus_b, lr=1e-2, weight_decay=1e-3)
    worst = {'grad': 0.0, 'm': 0.0, 'v': 0.0, 'param': 0.0, 'norm': 0.0}
    g = torch.Generator().manual_seed(2)
    for it in range(3):
        x = torch.randn(8, 3, 6, 6, generator=g)
        ref_b.zero_grad()
        ref_m(x).square().mean().backward()
        ref_b.finish()                                  # gather mode: gradients move into the flat buffers here
        with torch.no_grad():
            for k in range(len(ref_b.buckets)):
                fus_b.buckets[k][0].copy_(ref_b.buckets[k][0])
                fus_b.flat_params[k].copy_(ref_b.flat_params[k])
                fus_o.exp_avg[k].copy_(ref_o.exp_avg[k])
                fus_o.exp_avg_sq[k].copy_(ref_o.exp_avg_sq[k])
            fus_o.step_t.copy_(ref_o.step_t)
        max_norm = 0.05 if it != 1 else 1e9
        parallel.FUSED_ADAM = False
        n_ref = float(ref_o.clip_and_step(max_norm))
        parallel.FUSED_ADAM = True
        n_fus = float(fus_o.clip_and_step(max_norm))
        worst['norm'] = max(worst['norm'], abs(n_fus - n_ref) / n_ref)
        for k in range(len(ref_b.buckets)):
            worst['grad'] = max(worst['grad'], rel(fus_b.buckets[k][0], ref_b.buckets[k][0]))
            worst['m'] = max(worst['m'], rel(fus_o.exp_avg[k], ref_o.exp_avg[k]))
            worst['v'] = max(worst['v'], rel(fus_o.exp_avg_sq[k], ref_o.exp_avg_sq[k]))
            worst['param'] = max(worst['param'], err(fus_b.flat_params[k], ref_b.flat_params[k]))
    worst['steps'] = fus_o.step_count
    worst['buckets'] = len(fus_b.buckets)
    return worst


def se_block(ops):
    from stp3_amd import ops_fused
    out = {}
    for mlp in (False, True):
        ops_fused._SE_MLP = mlp
        torch.manual_seed(0)
        worst = 0.0
        # (S a multiple of 4: the batched gate kernels; 9 and 14: the general ones; 1100 channels: two chunks per thread, ragged)
        for n, c, s, hh, ww in [(5, 48, 12, 6, 7), (3, 200, 9, 4, 4), (2, 16, 4, 1, 3), (3, 960, 40, 2, 3), (2, 672, 28, 3, 2),
                                (2, 1100, 8, 2, 2), (2, 336, 14, 2, 2), (2, 1048, 64, 1, 2)]:
            x = torch.randn(n, c, hh, ww).contiguous(memory_format=torch.channels_last).requires_grad_()
            params = [(torch.randn(s, c, 1, 1) * 0.3).requires_grad_(), torch.randn(s).requires_grad_(),
                      (torch.randn(c, s, 1, 1) * 0.3).requires_grad_(), torch.randn(c).requires_grad_()]
            y = ops_fused._SeBlock.apply(x, *params)
            gy = torch.randn_like(y)
            y.backward(gy)
            got = [y.detach()] + [t.grad.clone() for t in [x] + params]
            for t in [x] + params:
                t.grad = None
            w1, b1, w2, b2 = params
            gate = torch.sigmoid(torch.nn.functional.silu(x.mean(dim=(2, 3)) @ w1.flatten(1).t() + b1) @ w2.flatten(1).t() + b2)
            ref = x * gate[:, :, None, None]
            ref.backward(gy)
            want = [ref.detach()] + [t.grad for t in [x] + params]
            worst = max([worst] + [rel(a, b) for a, b in zip(got, want)])
        out['mlp_kernels' if mlp else 'torch_mlp'] = worst
    return out


def bn_act(ops):
    from stp3_amd.layers import fused
    out = {}
    torch.manual_seed(1)
    for training in (True, False):
        for dtype, tag in ((torch.float32, 'f32'), (torch.bfloat16, 'bf16')):
            worst = 0.0
            for act, res_mode, with_sbias, with_oscale in [(ops.ACT_RELU, ops.RES_NONE, False, False),
                                                           (ops.ACT_SWISH, ops.RES_BEFORE_ACT, False, False),
                                                           (ops.ACT_NONE, ops.RES_AFTER_ACT, False, True),
                                                           (ops.ACT_RELU, ops.RES_NONE, True, False)]:
                n, c, h, w = 4, 24, 5, 9
                bn_a, bn_b = torch.nn.BatchNorm2d(c, eps=1e-3), torch.nn.BatchNorm2d(c, eps=1e-3)
                with torch.no_grad():
                    bn_a.weight.uniform_(0.5, 1.5); bn_a.bias.normal_(0, 0.2)
                    bn_a.running_mean.normal_(0, 0.2); bn_a.running_var.uniform_(0.5, 1.5)
                bn_b.load_state_dict(bn_a.state_dict())
                bn_a.train(training); bn_b.train(training)
                x0 = torch.randn(n, c, h, w).to(dtype).contiguous(memory_format=torch.channels_last)
                r0 = torch.randn(n, c, h, w).to(dtype).contiguous(memory_format=torch.channels_last) if res_mode else None
                sb0 = torch.randn(n, c) * 0.3 if with_sbias else None
                osc = torch.rand(n) + 0.5 if with_oscale else None
                gy = torch.randn(n, c, h, w).to(dtype).contiguous(memory_format=torch.channels_last)
                res = []
                for kernel, bn in ((True, bn_a), (False, bn_b)):
                    # the torch statement always runs in float32 on the same (bf16-representable) inputs: the kernels
                    # compute in float32 internally, a bf16 reference would round x + sbias first and flip ReLU masks
                    cast = (lambda t: t.clone()) if kernel else (lambda t: t.float())
                    x = cast(x0).requires_grad_()
                    r = cast(r0).requires_grad_() if r0 is not None else None
                    sb = sb0.clone().requires_grad_() if sb0 is not None else None
                    fn = fused.bn_act if kernel else fused.bn_act_reference
                    y = fn(bn, x, act, r, res_mode, sb, osc)
                    y.backward(gy if kernel else gy.float())
                    res.append([y.detach().float(), x.grad.float(), bn.weight.grad, bn.bias.grad] +
                               ([r.grad.float()] if r is not None else []) + ([sb.grad] if sb is not None else []) +
                               [bn.running_mean.clone(), bn.running_var.clone()])
                worst = max([worst] + [rel(a, b) for a, b in zip(*res)])
            out[f'{"train" if training else "eval"}_{tag}'] = worst
    return out


def bn_act_padded(ops):
    """35-channel BatchNorm layers in 40-lane rows (stp3_bn_dims.cpad): the padding lanes of every input hold NaN, the
    results must equal the float32 torch statement on the 35 real channels and be EXACTLY zero in the padding lanes."""
    from stp3_amd.layers import fused
    out = {}
    torch.manual_seed(11)
    n, c, cp, h, w = 3, 35, 40, 6, 7
    for training in (True, False):
        for dtype, tag in ((torch.float32, 'f32'), (torch.bfloat16, 'bf16')):
            worst, pad_clean = 0.0, True
            for act, res_mode, with_sbias in [(ops.ACT_RELU, ops.RES_NONE, True), (ops.ACT_SWISH, ops.RES_BEFORE_ACT, False),
                                              (ops.ACT_RELU, ops.RES_AFTER_ACT, False)]:
                bn_a, bn_b = torch.nn.BatchNorm2d(c, eps=1e-3), torch.nn.BatchNorm2d(c, eps=1e-3)
                with torch.no_grad():
                    bn_a.weight.uniform_(0.5, 1.5); bn_a.bias.normal_(0, 0.2)
                    bn_a.running_mean.normal_(0, 0.2); bn_a.running_var.uniform_(0.5, 1.5)
                bn_b.load_state_dict(bn_a.state_dict())
                bn_a.train(training); bn_b.train(training)

                def padded(t):
                    full = torch.full((n, cp, h, w), float('nan'), dtype=dtype).contiguous(memory_format=torch.channels_last)
                    full[:, :c] = t
                    return full
                x0 = torch.randn(n, c, h, w).to(dtype)
                r0 = torch.randn(n, c, h, w).to(dtype) if res_mode else None
                gy = torch.randn(n, c, h, w).to(dtype)
                sb0 = torch.randn(n, c) * 0.3 if with_sbias else None
                xk = padded(x0).requires_grad_()
                rk = padded(r0).requires_grad_() if r0 is not None else None
                sbk = sb0.clone().requires_grad_() if sb0 is not None else None
                yk = fused.bn_act(bn_a, xk, act, rk, res_mode, sbk)
                yk.backward(padded(gy))
                xr = x0.float().requires_grad_()
                rr = r0.float().requires_grad_() if r0 is not None else None
                sbr = sb0.clone().requires_grad_() if sb0 is not None else None
                yr = fused.bn_act_reference(bn_b, xr, act, rr, res_mode, sbr)
                yr.backward(gy.float())
                got = [yk.detach()[:, :c].float(), xk.grad[:, :c].float(), bn_a.weight.grad, bn_a.bias.grad,
                       bn_a.running_mean, bn_a.running_var]
                want = [yr.detach(), xr.grad, bn_b.weight.grad, bn_b.bias.grad, bn_b.running_mean, bn_b.running_var]
                if rr is not None:
                    got.append(rk.grad[:, :c].float()); want.append(rr.grad)
                if sbr is not None:
                    got.append(sbk.grad); want.append(sbr.grad)
                worst = max([worst] + [rel(a, b) for a, b in zip(got, want)])
                pads = [yk.detach()[:, c:], xk.grad[:, c:]]
                if rk is not None and res_mode == ops.RES_BEFORE_ACT:
                    pads.append(rk.grad[:, c:])
                pad_clean = pad_clean and all(bool((t == 0).all()) for t in pads) and tuple(yk.shape) == (n, cp, h, w)
            out[f'{"train" if training else "eval"}_{tag}'] = worst
            out[f'{"train" if training else "eval"}_{tag}_pad_zero'] = pad_clean
    return out


def causal_pair(ops):
    """stp3_causal_pair_fwd / _bwd against the torch construction (zero frame, two concatenations) -- copies and one
    two-term addition: bit-exact."""
    out = {}
    torch.manual_seed(12)
    b, t, h, w = 2, 3, 5, 6
    for dtype, c, tag in ((torch.bfloat16, 40, 'bf16'), (torch.float32, 12, 'f32'), (torch.bfloat16, 8, 'bf16_c8')):
        x0 = torch.randn(b * t, c, h, w).to(dtype).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(b * t, 2 * c, h, w).to(dtype).contiguous(memory_format=torch.channels_last)
        xk = x0.clone().requires_grad_()
        yk = ops.causal_pair(xk, t)
        yk.backward(gy)
        xr = x0.clone().requires_grad_()
        x5 = xr.view(b, t, c, h, w)
        prev = torch.cat([torch.zeros_like(x5[:, :1]), x5[:, :-1]], dim=1).view(b * t, c, h, w)
        yr = torch.cat([prev, xr], dim=1)
        yr.backward(gy)
        out[tag] = bool(torch.equal(yk.detach(), yr.detach()) and torch.equal(xk.grad, xr.grad))
    xs = torch.randn(b * t, 24, h, w).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)[:, :16]
    out['strided'] = bool(torch.equal(ops.causal_pair(xs, t)[:, 16:], xs))     # a channel slice read in place (ldx = 24)
    return out


def upsample(ops):
    """stp3_upsample_bilinear_fwd / _bwd against F.interpolate (float32) and its autograd."""
    import torch.nn.functional as F
    out = {}
    torch.manual_seed(13)
    for tag, dtype, c, h, w, scale in (('f32_x2', torch.float32, 8, 5, 7, 2), ('bf16_x2', torch.bfloat16, 16, 6, 5, 2),
                                       ('bf16_x3', torch.bfloat16, 8, 4, 3, 3), ('f32_x4', torch.float32, 4, 3, 1, 4),
                                       ('bf16_1x1', torch.bfloat16, 8, 1, 1, 2)):
        x0 = torch.randn(2, c, h, w).to(dtype).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(2, c, h * scale, w * scale).to(dtype).contiguous(memory_format=torch.channels_last)
        xk = x0.clone().requires_grad_()
        yk = ops.upsample_bilinear(xk, scale)
        yk.backward(gy)
        xr = x0.float().requires_grad_()
        yr = F.interpolate(xr, scale_factor=scale, mode='bilinear', align_corners=False)
        yr.backward(gy.float())
        out[tag] = {'y': rel(yk.detach().float(), yr.detach().to(dtype).float()),
                    'dx': rel(xk.grad.float(), xr.grad.to(dtype).float()),
                    'shape': list(yk.shape) == list(yr.shape) and yk.dtype == dtype}
    # channel slices: x read with its own row stride, the gradient of a concatenation read in place
    wide = torch.randn(2, 24, 4, 5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xs = wide[:, 8:24].detach().requires_grad_()
    skip = torch.randn(2, 8, 8, 10).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    cat = torch.cat([skip, ops.upsample_bilinear(xs, 2)], dim=1)
    g = torch.randn_like(cat)
    cat.backward(g)
    xr = wide[:, 8:24].float().requires_grad_()
    F.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=False).backward(g[:, 8:].float())
    out['sliced'] = {'y': rel(cat[:, 8:].detach().float(), F.interpolate(xr.detach(), scale_factor=2, mode='bilinear',
                                                                         align_corners=False).to(torch.bfloat16).float()),
                     'dx': rel(xs.grad.float(), xr.grad.to(torch.bfloat16).float()), 'shape': True}
    return out


def conv(ops):
    import torch.nn.functional as F
    out = {}
    torch.manual_seed(2)
    for name, (cin, cout, k, s, p, d, bias) in {'3x3': (16, 24, 3, 1, 1, 1, True), '1x1': (8, 40, 1, 1, 0, 1, False),
                                                 '3x3s2': (16, 16, 3, 2, 1, 1, False), 'dil2': (8, 8, 3, 1, 2, 2, False),
                                                 # Cin == 8: the weight gradient folds the taps into its ci tile (one / two tap groups)
                                                 'stem': (8, 48, 3, 2, 1, 1, False), 'c8_5x5': (8, 24, 5, 1, 2, 1, True),
                                                 'c8_2x2taps': (8, 72, 3, 1, 1, 1, False),
                                                 # strided layers: the data gradient runs per input phase
                                                 '7x7s2': (16, 8, 7, 2, 3, 1, False), '1x1s2': (8, 16, 1, 2, 0, 1, False),
                                                 '3x3s2p0': (8, 8, 3, 2, 0, 1, True), '3x3s3': (8, 8, 3, 3, 1, 1, False),
                                                 '5x5s2': (8, 16, 5, 2, 2, 1, False),
                                                 # weight gradient with several taps per workgroup: a kernel row of a wide
                                                 # layer (128 x 64 and 64 x 128 tiles), several ci / co tiles
                                                 '3x3_co96': (40, 96, 3, 1, 1, 1, False), '3x3_ci96': (96, 40, 3, 1, 1, 1, False),
                                                 '3x3_d2_co136': (72, 136, 3, 1, 2, 2, False),
                                                 '3x3_wide': (16, 24, 3, 1, 1, 1, False), '3x3s2_wide': (8, 16, 3, 2, 1, 1, False),
                                                 # enough pixels that the forward / data-gradient kernel takes its 128 x 128 tile (the
                                                 # launcher keeps the 128 x 64 tile below 512 workgroups): two channel tiles, the second ragged
                                                 '3x3_tile128': (24, 136, 3, 1, 1, 1, False)}.items():
        # '*_wide': rows longer than one 64-pixel weight-gradient step (the incremental pixel coordinates carry rarely)
        shape = (1, cin, 5, 150) if name.endswith('_wide') else (1, cin, 128, 256) if name.endswith('_tile128') else (2, cin, 9, 12)
        x0 = torch.randn(*shape).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w0 = (torch.randn(cout, cin, k, k) * 0.2).to(torch.bfloat16).float()       # bf16-representable weights
        b0 = torch.randn(cout) if bias else None
        x = x0.clone().requires_grad_()
        w = w0.clone().requires_grad_()
        b = b0.clone().requires_grad_() if bias else None
        y = ops.conv2d(x, w, b, s, p, d, out_dtype=torch.float32)
        gy = torch.randn_like(y).to(torch.bfloat16).float()
        y.backward(gy)
        xr = x0.float().requires_grad_()
        wr = w0.clone().requires_grad_()
        br = b0.clone().requires_grad_() if bias else None
        yr = F.conv2d(xr, wr, br, s, p, d)
        yr.backward(gy)
        out[name] = {'y': rel(y.detach(), yr.detach()), 'dx': rel(x.grad.float(), xr.grad), 'dw': rel(w.grad, wr.grad),
                     'db': rel(b.grad, br.grad) if bias else 0.0}
        if s > 1 and d == 1:
            # the per-phase route of the data gradient (ops.conv2d_data_grad takes it for big layers only)
            wt = w0.to(torch.bfloat16).flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
            gyb = gy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            dxp = ops._strided_dgrad(gyb, wt, tuple(x0.shape), s, (p, p), {})
            out[name]['dx_phases'] = rel(dxp.float(), xr.grad) if dxp is not None else -1.0
    return out


def wgrad_defer(ops):
    """Weight gradients written into their bucket slice, their split-K sums deferred to ONE launch per backward pass
    (ops.DEFER_WGRAD_REDUCE; stp3_conv2d_wgrad_partials + stp3_conv2d_wgrad_reduce_batch): the flat buckets against the plain
    route (fresh tensors, reduced at once, gathered), bit for bit.  One weight is applied three times (must NOT defer), one
    has padded output channels (takes the gather route), and a second pass meets the state the first one left."""
    import torch.nn as nn
    from stp3_amd.parallel import GradientBuckets

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Conv2d(16, 32, 3, padding=1, bias=True)
            self.shared = nn.Conv2d(32, 32, 3, padding=1, bias=False)
            self.odd = nn.Conv2d(32, 35, 1, bias=False)
            self.tail = nn.Conv2d(32, 72, 1, bias=False)

        def forward(self, x):
            h = ops.conv2d(x, self.a.weight, self.a.bias, 1, 1, 1)
            for _ in range(3):
                h = torch.relu(ops.conv2d(h, self.shared.weight, None, 1, 1, 1))
            return ops.conv2d(h, self.odd.weight, None, 1, 0, 1).float().square().mean() + \
                ops.conv2d(h, self.tail.weight, None, 1, 0, 1).float().square().mean()

    def run(direct, defer):
        ops.DIRECT_BUCKET_GRADS, ops.DEFER_WGRAD_REDUCE = direct, defer
        torch.manual_seed(3)
        model = Net().to(memory_format=torch.channels_last)         # the layout of dw: what the product's models hold
        buckets = GradientBuckets(model, bucket_bytes=32 << 10)
        x = torch.randn(2, 16, 9, 20, generator=torch.Generator().manual_seed(9)).to(torch.bfloat16)
        x = x.contiguous(memory_format=torch.channels_last)
        flats, pending = [], []
        for _ in range(2):
            buckets.zero_grad()
            model(x).backward()
            pending.append(ops.pending_wgrad_reductions())
            buckets.finish()
            flats.append(torch.cat([f.clone() for f, _ in buckets.buckets]))
        return flats, pending

    keep = ops.DIRECT_BUCKET_GRADS, ops.DEFER_WGRAD_REDUCE
    plain, p0 = run(False, False)
    direct, p1 = run(True, False)
    deferred, p2 = run(True, True)
    ops.DIRECT_BUCKET_GRADS, ops.DEFER_WGRAD_REDUCE = keep
    return {'direct_equal': all(torch.equal(a, b) for a, b in zip(plain, direct)),
            'deferred_equal': all(torch.equal(a, b) for a, b in zip(plain, deferred)),
            'nonzero': float(plain[0].abs().max()) > 0, 'pending': [p0, p1, p2], 'left_over': ops.pending_wgrad_reductions()}


def assembled_weights(ops):
    """ops.ASSEMBLED_WEIGHTS: the weights the model puts together from parameters (padded channel lanes and causal taps of the
    temporal block, the heads of the decoder merged, ASPP's kept taps and split projection, the padded stem) as shadows written
    by stp3_conv2d_prep_weights and gradients cut back by stp3_conv2d_scatter_weight_grads -- against the same modules building
    those weights with torch (pad / cat / block_diag / slice and their backward).  Same kernels, same bf16 operands: outputs and
    flat gradient buckets must agree to the rounding of one bf16 sum; two passes with an optimizer-like update in between (the
    shadows must follow the parameters), then the entries' use counters and pending lists must be clean."""
    import torch.nn as nn
    from stp3_amd.layers import temporal as T, convolutions as C, fused
    from stp3_amd.models import decoder as D, efficientnet as E
    from stp3_amd.parallel import GradientBuckets
    torch.is_autocast_enabled = lambda *a: True
    torch.get_autocast_gpu_dtype = lambda: torch.bfloat16
    torch.get_autocast_dtype = lambda *a: torch.bfloat16
    gate = {'perceive_hdmap': False, 'predict_pedestrian': True, 'predict_instance': True, 'predict_future_flow': False, 'planning': False}

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = E.StaticSamePadConv2d(3, 16, 3, image_size=32, stride=2)
            self.first = T.TemporalBlock(22, 16, use_pyramid_pooling=True, pool_sizes=[(2, 16, 24)])   # 16 planes + 6 constants
            self.second = T.TemporalBlock(16, 16)
            self.aspp = C.ASPP(16, [2, 18, 40], 16)          # 16 x 24 map: all taps, centre row only, centre tap only
            self.decoder = D.Decoder(16, 2, 2, 2, gate)

        def forward(self, img, extra):
            b, t = extra.shape[0], extra.shape[2]
            f = self.stem(img)                                                   # (b*t, 16, 16, 24)
            x = f.view(b, t, *f.shape[1:]).permute(0, 2, 1, 3, 4)
            x = self.second(self.first(x, extra))
            y = x.permute(0, 2, 1, 3, 4).reshape(b * t, *x.shape[1:2], *x.shape[3:])
            y = self.aspp(y.contiguous(memory_format=torch.channels_last))
            out = self.decoder(y.view(b, t, *y.shape[1:]))
            return sum(v.float().square().mean() for v in out.values() if v is not None) + f.float().mean()

    def run(on):
        ops.ASSEMBLED_WEIGHTS = on
        torch.manual_seed(5)
        from stp3_amd.utils import to_channels_last
        model = to_channels_last(Net()).train()
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        buckets = GradientBuckets(model, bucket_bytes=64 << 10)
        g = torch.Generator().manual_seed(2)
        img = torch.randn(4, 3, 32, 48, generator=g)
        extra = torch.randn(2, 6, 2, generator=g)
        losses, flats, pending, calls = [], [], [], []
        for it in range(2):
            buckets.zero_grad()
            with torch.autocast('cpu', dtype=torch.bfloat16):
                loss = model(img, extra)
            loss.backward()
            pending.append(ops.pending_wgrad_reductions())
            buckets.finish()
            fused.flush_batch_counters()
            losses.append(float(loss))
            flats.append(torch.cat([f.clone() for f, _ in buckets.buckets]))
            with torch.no_grad():                                                # "optimizer": through the flat buffers
                for fp in buckets.flat_params:
                    fp.mul_(0.97)
            ops.invalidate_weight_cache()
        n_asm = sum(len(t.assembled) for t in ops._SHADOW_TABLES.values())
        return losses, flats, pending, n_asm

    keep = ops.ASSEMBLED_WEIGHTS
    plain = run(False)
    for t in ops._SHADOW_TABLES.values():
        assert not t.assembled
    asm = run(True)
    ops.ASSEMBLED_WEIGHTS = keep
    scale = [float(f.abs().max()) for f in plain[1]]
    return {'loss': max(abs(a - b) / abs(b) for a, b in zip(asm[0], plain[0])),
            'grads': max(float((a - b).abs().max()) / s for a, b, s in zip(asm[1], plain[1], scale)),
            'grads_l2': max(float((a - b).norm() / b.norm()) for a, b in zip(asm[1], plain[1])),
            'nonzero': min(scale) > 0, 'assembled': asm[3], 'pending_before_finish': asm[2],
            'left_over': ops.pending_wgrad_reductions(),
            'uses_reset': all(e['uses'] <= 1 for t in ops._SHADOW_TABLES.values() for e in t.assembled.values())}


def small_linear(ops):
    """stp3_linear_fwd / _bwd (the pooled descriptors' 1x1 convolutions as one launch each way) against float64 torch: output,
    input / weight / bias gradients; with and without a bias, row counts that are not multiples of the 16-lane split."""
    import torch.nn.functional as F
    out = {}
    torch.manual_seed(0)
    for name, (m, k, n, bias) in {'m12_k64_n128_b': (12, 64, 128, True), 'm16_k70_n23': (16, 70, 23, False),
                                  'm72_k160_n64_b': (72, 160, 64, True), 'm3_k6_n35': (3, 6, 35, False)}.items():
        x, w = torch.randn(m, k, requires_grad=True), torch.randn(n, k, requires_grad=True)
        b = torch.randn(n, requires_grad=True) if bias else None
        assert ops.small_linear_supported(x, w, b)
        y = ops.small_linear(x, w, b)
        gy = torch.randn(m, n)
        y.backward(gy)
        xr, wr = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
        br = b.detach().double().requires_grad_() if bias else None
        F.linear(xr, wr, br).backward(gy.double())
        out[name] = {'y': rel(y.detach(), F.linear(xr, wr, br).detach()), 'dx': rel(x.grad, xr.grad), 'dw': rel(w.grad, wr.grad),
                     'db': rel(b.grad, br.grad) if bias else 0.0}
    # the weight as a run of COLUMNS of a wider matrix, read in place (row stride 70)
    x, wide = torch.randn(12, 6, requires_grad=True), torch.randn(35, 70, requires_grad=True)
    y = ops.small_linear(x, wide[:, 64:], None)
    gy = torch.randn(12, 35)
    y.backward(gy)
    xr, wr = x.detach().double().requires_grad_(), wide.detach().double().requires_grad_()
    F.linear(xr, wr[:, 64:]).backward(gy.double())
    out['columns_of_a_wider_matrix'] = {'y': rel(y.detach(), F.linear(xr, wr[:, 64:]).detach()), 'dx': rel(x.grad, xr.grad),
                                        'dw': rel(wide.grad, wr.grad), 'db': 0.0}
    return out


def fan_out(ops):
    """ops.fan_out: the n gradients of a tensor with n consumers added in one pass (stp3_sum_n) -- against the sum in
    float64; bf16 (float32 accumulation, one rounding: at least as close as the pairwise bf16 additions of autograd) and
    float32, vector and scalar paths, odd sizes."""
    out = {}
    torch.manual_seed(21)
    for name, (shape, dt, n) in {'bf16_5': ((3, 24, 5, 7), torch.bfloat16, 5), 'f32_3': ((2, 8, 6, 6), torch.float32, 3),
                                 'bf16_odd_6': ((1, 3, 5, 7), torch.bfloat16, 6), 'f32_8': ((2, 5, 3, 3), torch.float32, 8),
                                 'bf16_cl_4': ((2, 16, 4, 6), torch.bfloat16, 4)}.items():
        x = torch.randn(*shape).to(dt)
        if 'cl' in name:
            x = x.contiguous(memory_format=torch.channels_last)
        x.requires_grad_()
        ws = [torch.randn(*shape).to(dt) for _ in range(n)]
        if 'cl' in name:
            ws = [w.contiguous(memory_format=torch.channels_last) for w in ws]
        handles = ops.fan_out(x, n)
        sum((h * w).sum() for h, w in zip(handles, ws)).backward()
        exact = sum(w.double() for w in ws)
        pair = ws[0].clone()
        for w in ws[1:]:
            pair = pair + w                                            # what autograd does: pairwise, rounding every time
        out[name] = {'err': rel(x.grad.double(), exact), 'pairwise_err': rel(pair.double(), exact),
                     'layout': list(x.grad.stride()) == list(x.stride()), 'dtype': str(x.grad.dtype)}
    # handles nobody consumed (the decoder hands out one per head and its merged operator takes a single one): they come
    # back as None, not as zero tensors -- the sum is that of the consumed ones, in one pass, in the tensor's layout
    x = torch.randn(2, 16, 4, 6).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    ws = [torch.randn(2, 16, 4, 6).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(2)]
    handles = ops.fan_out(x, 7)
    ((handles[1] * ws[0]).sum() + (handles[5] * ws[1]).sum()).backward()
    exact = ws[0].double() + ws[1].double()
    out['bf16_unused_handles'] = {'err': rel(x.grad.double(), exact), 'pairwise_err': rel((ws[0] + ws[1]).double(), exact),
                                  'layout': list(x.grad.stride()) == list(x.stride()), 'dtype': str(x.grad.dtype)}
    # one consumer is a whole-plane mean (layers/fused.plane_mean): its gradient is constant over every (sample, channel) plane
    # and joins the single-pass sum as a broadcast addend (stp3_sum_n_plane) instead of a materialised tensor
    from stp3_amd.layers import fused
    for name, dt in (('bf16_plane', torch.bfloat16), ('f32_plane', torch.float32)):
        x = torch.randn(3, 16, 5, 7).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_()
        ws = [torch.randn(3, 16, 5, 7).to(dt).contiguous(memory_format=torch.channels_last) for _ in range(2)]
        pw = torch.randn(3, 16)
        handles = ops.fan_out(x, 3)
        ((handles[0] * ws[0]).sum() + (fused.plane_mean(handles[1]) * pw).sum() + (handles[2] * ws[1]).sum()).backward()
        plane = (pw / 35.0).to(dt).double()[:, :, None, None].expand(3, 16, 5, 7)
        exact = ws[0].double() + ws[1].double() + plane
        out[name] = {'err': rel(x.grad.double(), exact), 'pairwise_err': rel(((ws[0] + ws[1]) + plane.to(dt)).double(), exact),
                     'layout': list(x.grad.stride()) == list(x.stride()), 'dtype': str(x.grad.dtype),
                     # the forward: the whole-plane mean itself (stp3_se_pool) against the float64 mean of the same values
                     'mean_err': rel(fused.plane_mean(x.detach()).double(), x.detach().double().mean((2, 3)))}
    return out


def conv_f32(ops):
    """FLOAT32 operands through the three-term bf16 split (ops.conv2d_f32) on the MFMA kernels, against the same
    convolution in FLOAT64: the route the float32 legs of the GPU parity tests take."""
    import torch.nn.functional as F
    out = {}
    torch.manual_seed(12)
    for name, (cin, cout, k, s, p, d, bias) in {'3x3': (16, 24, 3, 1, 1, 1, True), '1x1_head': (8, 2, 1, 1, 0, 1, True),
                                                 '7x7s2': (16, 8, 7, 2, 3, 1, False), 'dil2': (8, 8, 3, 1, 2, 2, False),
                                                 'stem': (8, 48, 3, 2, 1, 1, False), '3x3s2': (16, 16, 3, 2, 1, 1, False),
                                                 '3x3_co96': (40, 96, 3, 1, 1, 1, False)}.items():
        x0 = torch.randn(2, cin, 9, 12) * 3.0 + 0.5                                  # plain NCHW float32, as the parity runs hand it over
        w0 = torch.randn(cout, cin, k, k) * 0.2
        b0 = torch.randn(cout) if bias else None
        x, w = x0.clone().requires_grad_(), w0.clone().requires_grad_()
        b = b0.clone().requires_grad_() if bias else None
        y = ops.conv2d_f32(x, w, b, s, p, d)
        gy = torch.randn_like(y)
        y.backward(gy)
        xr, wr = x0.double().requires_grad_(), w0.double().requires_grad_()
        br = b0.double().requires_grad_() if bias else None
        yr = F.conv2d(xr, wr, br, s, p, d)
        yr.backward(gy.double())
        # what plain float32 arithmetic makes of the same convolution, for scale
        y32 = F.conv2d(x0, w0, b0, s, p, d)
        out[name] = {'y': rel(y.detach(), yr.detach()), 'dx': rel(x.grad, xr.grad), 'dw': rel(w.grad, wr.grad),
                     'db': rel(b.grad, br.grad) if bias else 0.0, 'torch_f32_y': rel(y32, yr.detach()),
                     'dtypes': [str(y.dtype), str(x.grad.dtype), str(w.grad.dtype)]}
    return out


def dwconv(ops):
    import torch.nn.functional as F
    out = {}
    torch.manual_seed(3)
    cases = {'k3s1': (3, 1, (1, 1, 1, 1), False), 'k5s2': (5, 2, (1, 2, 1, 2), False), 'k3s2': (3, 2, (0, 1, 0, 1), False),
             'k7s1_bias': (7, 1, (3, 3, 3, 3), True)}                  # the ConvNeXt blocks' 7x7 layer (has a bias)
    for name, (k, s, pad, bias) in cases.items():
        c = 24
        x0 = torch.randn(2, c, 9, 12).contiguous(memory_format=torch.channels_last)
        w0 = torch.randn(c, 1, k, k) * 0.3
        b0 = torch.randn(c) * 0.5 if bias else None
        x, w = x0.clone().requires_grad_(), w0.clone().requires_grad_()
        b = b0.clone().requires_grad_() if bias else None
        y = ops.depthwise_conv2d(x, w, s, pad, bias=b)
        gy = torch.randn_like(y)
        y.backward(gy)
        xr, wr = x0.clone().requires_grad_(), w0.clone().requires_grad_()
        br = b0.clone().requires_grad_() if bias else None
        yr = F.conv2d(F.pad(xr, pad), wr, br, s, 0, 1, c)
        yr.backward(gy)
        out[name] = {'y': rel(y.detach(), yr.detach()), 'dx': rel(x.grad, xr.grad), 'dw': rel(w.grad, wr.grad)}
        if bias:
            out[name]['db'] = rel(b.grad, br.grad)
    return out


def mbconv_mid(ops):
    """Depthwise -> BatchNorm -> swish -> squeeze-excite as ONE operator (ops_fused.dw_bn_se: statistics in the depthwise
    epilogue, swish(BN(.)) never written, one backward pass for the gate gradient and the BatchNorm reductions) against
    float32 torch autograd on the same (bf16-representable) data: outputs, input gradient and every parameter gradient,
    running statistics; stride 1 and 2, kernel 3 and 5, float32 and bf16 activations."""
    import torch.nn as nn
    import torch.nn.functional as F
    from stp3_amd import ops_fused
    from stp3_amd.models.efficientnet import StaticSamePadConv2d
    out = {}
    cases = {'k3s1_c24': (24, 3, 1, 3, 9, 12, 8), 'k5s2_c48': (48, 5, 2, 2, 11, 13, 24), 'k3s2_c16': (16, 3, 2, 5, 8, 8, 8),
             'k5s1_c272': (272, 5, 1, 2, 6, 5, 8)}
    for name, (c, k, stride, n, h, w, img) in cases.items():
        for dtype, tag in ((torch.float32, 'f32'), (torch.bfloat16, 'bf16')):
            g = torch.Generator().manual_seed(11)
            cl = torch.channels_last
            x0 = torch.randn(n, c, h, w, generator=g).to(dtype).contiguous(memory_format=cl)
            s = max(1, c // 4)
            res = []
            for mode in ('fused', 'unmerged', 'torch'):
                # 'unmerged': the stand-alone reductions between the passes (what more than one rank runs) -- the merged form
                # must give the same BITS (ops_fused.MERGE_SMALL_REDUCTIONS)
                ops_fused.MERGE_SMALL_REDUCTIONS = mode != 'unmerged'
                dw = StaticSamePadConv2d(c, c, k, img, stride=stride, groups=c)
                bn = nn.BatchNorm2d(c, momentum=0.01, eps=1e-3)
                r1, r2 = StaticSamePadConv2d(c, s, 1, 1, bias=True), StaticSamePadConv2d(s, c, 1, 1, bias=True)
                gp = torch.Generator().manual_seed(5)
                with torch.no_grad():
                    dw.weight.copy_(torch.randn(dw.weight.shape, generator=gp) * 0.3)
                    bn.weight.copy_(torch.rand(c, generator=gp) + 0.5); bn.bias.copy_(torch.randn(c, generator=gp) * 0.2)
                    r1.weight.copy_(torch.randn(r1.weight.shape, generator=gp) * 0.3); r1.bias.copy_(torch.randn(s, generator=gp) * 0.2)
                    r2.weight.copy_(torch.randn(r2.weight.shape, generator=gp) * 0.3); r2.bias.copy_(torch.randn(c, generator=gp) * 0.2)
                if mode != 'torch':
                    x = x0.clone().requires_grad_()
                    assert ops_fused.dw_bn_se_supported(x, dw, bn)
                    y = ops_fused.dw_bn_se(x, dw, bn, r1, r2, group=False)
                else:
                    x = x0.float().requires_grad_()
                    e2 = F.conv2d(F.pad(x, dw._pad), dw.weight, None, stride, 0, 1, c)
                    if dtype == torch.bfloat16:
                        e2 = e2 + (e2.to(torch.bfloat16).float() - e2).detach()        # the kernel stores E2 in bf16
                    sact = F.silu(bn(e2))
                    gate = torch.sigmoid(F.linear(F.silu(F.linear(sact.mean((2, 3)), r1.weight.flatten(1), r1.bias)),
                                                  r2.weight.flatten(1), r2.bias))
                    y = sact * gate[:, :, None, None]
                if mode == 'fused':
                    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(dtype).contiguous(memory_format=cl)
                y.backward(gy if mode != 'torch' else gy.float())
                res.append([y.detach().float(), x.grad.float(), dw.weight.grad, bn.weight.grad, bn.bias.grad, r1.weight.grad,
                            r1.bias.grad, r2.weight.grad, r2.bias.grad, bn.running_mean.clone(), bn.running_var.clone()])
            ops_fused.MERGE_SMALL_REDUCTIONS = True
            names = ['y', 'dx', 'ddw', 'dgamma', 'dbeta', 'dw1', 'db1', 'dw2', 'db2', 'rmean', 'rvar']
            out[f'{name}_{tag}'] = {k_: rel(a, b) for k_, a, b in zip(names, res[0], res[2])}
            out[f'{name}_{tag}']['merged_differs'] = float(sum(0 if torch.equal(a, b) else 1 for a, b in zip(res[0], res[1])))
    return out


def losses(ops):
    """csrc/stp3_loss.hip through stp3_amd.losses (GPU route) against the torch statements of the same module (CPU
    route): values and gradients of the segmentation (top-k + future discount + class weights + ignore), HD-map
    (two elements, one with top-k), depth (48 classes) and regression (L1 / L2, ignore mask) losses, float32 and bf16
    logits in channels-last and contiguous memory; and the nearest label warp against F.grid_sample."""
    import torch.nn.functional as F
    from stp3_amd import losses as L, geometry as geo, ops_loss
    out = {}
    g = torch.Generator().manual_seed(9)
    is_cuda = torch.Tensor.is_cuda

    def both(make_loss, pred0, *args):
        res = []
        for kernel in (True, False):
            torch.Tensor.is_cuda = is_cuda if kernel else property(lambda self: False)
            p = (pred0.detach().clone() if kernel else pred0.detach().float().clone()).requires_grad_()
            v = make_loss()(p, *args)
            v.backward()
            res.append((v.detach().double(), p.grad.double()))
        torch.Tensor.is_cuda = is_cuda
        return {'value': rel(res[0][0], res[1][0]), 'grad': rel(res[0][1], res[1][1])}

    b, s_, h, w = 2, 3, 12, 17
    seg = (torch.rand(b, s_, 1, h, w, generator=g) < 0.2).long()
    seg[0, 1, 0, :2] = 255                                        # ignored pixels
    for tag, dtype, cl in (('f32_nchw', torch.float32, False), ('bf16_nhwc', torch.bfloat16, True)):
        pred = (torch.randn(b * s_, 2, h, w, generator=g) * 2).to(dtype)
        if cl:
            pred = pred.contiguous(memory_format=torch.channels_last)
        pred = pred.view(b, s_, 2, h, w)
        out[f'seg_topk_{tag}'] = both(lambda: L.SegmentationLoss(torch.Tensor([1.0, 2.0]), use_top_k=True, top_k_ratio=0.25,
                                                                 future_discount=0.95), pred, seg, 2)
        out[f'seg_all_{tag}'] = both(lambda: L.SegmentationLoss(torch.Tensor([1.0, 2.0]), use_top_k=False), pred, seg, 3)
    # many exact ties at the threshold: logits on a coarse grid -> identical losses; value exact, gradient mass shared
    predq = (torch.randint(-2, 3, (b * s_, 2, h, w), generator=g).float() * 0.5).view(b, s_, 2, h, w)
    r = both(lambda: L.SegmentationLoss(torch.Tensor([1.0, 2.0]), use_top_k=True, top_k_ratio=0.25), predq, seg, 3)
    out['seg_topk_ties'] = {'value': r['value']}
    # rows at and beyond what the selection kernel keeps in registers (40 960 losses): 200 x 200 (the BEV) and 210 x 200
    for tag, (hh, ww) in (('bev_row', (200, 200)), ('long_row', (210, 200))):
        segl = (torch.rand(1, 1, 1, hh, ww, generator=g) < 0.2).long()
        predl = torch.randn(1, 1, 2, hh, ww, generator=g) * 2
        out[f'seg_topk_{tag}'] = both(lambda: L.SegmentationLoss(torch.Tensor([1.0, 2.0]), use_top_k=True, top_k_ratio=0.25), predl, segl, 1)
    hd = (torch.rand(b, 2, h, w, generator=g) < 0.3).long()
    predh = torch.randn(b, 4, h, w, generator=g).contiguous(memory_format=torch.channels_last)
    out['hdmap'] = both(lambda: L.HDmapLoss(torch.Tensor([[1.0, 5.0], [1.0, 1.0]]), [1, 2], [True, False], [0.25, 0.25]), predh, hd)
    dep = torch.randint(0, 48, (1, 2, 3, 7, 9), generator=g)
    predd = torch.randn(1, 2, 3, 48, 7, 9, generator=g) * 3
    out['depth'] = both(lambda: L.DepthLoss(), predd, dep)
    tgt = torch.randn(b, s_, 2, h, w, generator=g)
    tgt[:, :, :, :3] = 255.0
    predr = torch.randn(b, s_, 2, h, w, generator=g)
    out['reg_l1'] = both(lambda: L.SpatialRegressionLoss(norm=1, future_discount=0.95, ignore_index=255), predr, tgt, 2)
    out['reg_l2'] = both(lambda: L.SpatialRegressionLoss(norm=2, future_discount=0.95, ignore_index=255), predr, tgt, 2)
    out['reg_all_ignored'] = {'value': float(L.SpatialRegressionLoss(norm=1)(predr, torch.full_like(tgt, 255.0), 2).abs())}
    # nearest warp
    x = torch.randint(0, 5, (6, 3, 20, 24), generator=g).float()
    ang = torch.randn(6, generator=g) * 0.1
    theta = torch.stack([torch.cos(ang), -torch.sin(ang), torch.randn(6, generator=g) * 0.1, torch.sin(ang), torch.cos(ang),
                         torch.randn(6, generator=g) * 0.1], dim=-1).view(6, 2, 3)
    y = ops_loss.warp_nearest(x, theta, [0, 0, 1, 0, 0, 1])
    ref = geo.warp_with_theta(x, theta, 'nearest')
    ref[2], ref[5] = x[2], x[5]
    out['warp'] = {'mismatch_fraction': float((y != ref).float().mean())}
    return out


def plan(ops):
    """csrc/stp3_plan.hip through stp3_amd.cost.Cost_Function (GPU route) against the torch statements of the same module
    (CPU route, bit-equal to the reference on the fixtures of tests/golden/planning.npz): both cost tensors and the
    cost-volume gradient, for label and logit hd maps, with and without a target point, and for a single (expert)
    trajectory."""
    from stp3_amd.config import perception_cfg
    from stp3_amd.cost import Cost_Function
    from tests import helpers as H
    cfg = perception_cfg(**{'N_FUTURE_FRAMES': 4, 'PLANNING.ENABLED': True, 'PLANNING.SAMPLE_NUM': 60})
    ins = H.planning_inputs(cfg)
    cf = Cost_Function(cfg)
    is_cuda = torch.Tensor.is_cuda
    out = {}
    for form in ('labels', 'logits', 'labels_no_target', 'expert'):
        hd = ins['hdmap_logits'] if form == 'logits' else ins['hdmap_labels']
        lane, drv = (hd[:, 0:1], hd[:, 1:2]) if hd.shape[1] == 2 else (hd[:, 0:2], hd[:, 2:4])
        tgt = torch.zeros_like(ins['target']) if form == 'labels_no_target' else ins['target']
        tr = ins['gt_trajs'][:, None, :, :2] if form == 'expert' else ins['trajs'][..., :2]
        w = ins['w_fo'][:, :tr.shape[1]]
        res = []
        for kernel in (True, False):
            torch.Tensor.is_cuda = is_cuda if kernel else property(lambda self: False)
            cv = ins['cost_volume'].clone().requires_grad_()
            fc, fo = cf(cv, tr.clone(), ins['occupancy'], lane.clone(), drv.clone(), tgt)
            (fo * w).sum().backward()
            res.append((fc.detach(), fo.detach(), cv.grad))
        torch.Tensor.is_cuda = is_cuda
        out[form] = {k: err(a, b) for k, a, b in zip(('cost_fc', 'cost_fo', 'd_cost_volume'), res[0], res[1])}
    return out


def labels(ops):
    """csrc/stp3_labels.hip through stp3_amd.datas (GPU route): polygon fill against the oracle's fixture and against the
    CPU statement on random polygons (3 .. 8 vertices, partly outside the image, painted over each other); instance labels
    against the REFERENCE's own function (tests/golden/labels.npz)."""
    from stp3_amd import datas
    from tests import helpers as H
    is_cuda = torch.Tensor.is_cuda
    g = H.load('labels.npz')
    polys = list(g['poly/vertices'])
    want = np.unpackbits(g['poly/oracle'], axis=1).reshape(40, 200, 200)
    got = datas.fill_polygons(polys, [1.0] * 40, list(range(40)), 40, (200, 200), device='cpu').numpy()
    rng = np.random.default_rng(5)
    many = [rng.integers(-20, 120, (int(rng.integers(3, 9)), 2)) for _ in range(60)]
    vals = [float(i % 7 + 1) for i in range(60)]
    idx = [i % 5 for i in range(60)]
    a = datas.fill_polygons(many, vals, idx, 5, (96, 104), device='cpu').numpy()
    # thin, degenerate and self-touching polygons on a tiny lattice: edges meet exactly at pixel centres all the time
    # (two sorted active edges AT one pixel are the pair [x, x] and paint it -- the closed form once missed that)
    thin = [rng.integers(0, 14, (int(rng.integers(3, 9)), 2)) for _ in range(300)]
    c = datas.fill_polygons(thin, [1.0] * 300, list(range(300)), 300, (16, 16), device='cpu').numpy()
    torch.Tensor.is_cuda = property(lambda self: False)
    b = datas.fill_polygons(many, vals, idx, 5, (96, 104)).numpy()
    d = datas.fill_polygons(thin, [1.0] * 300, list(range(300)), 300, (16, 16)).numpy()
    torch.Tensor.is_cuda = is_cuda
    inst = torch.from_numpy(g['instance/ids'].astype(np.int64))
    ego = torch.from_numpy(g['instance/future_egomotion'])
    center, offset, flow = datas.instance_labels(inst, ego, int(g['instance/num_instances'][0]), spatial_extent=(50.0, 50.0))
    return {'fixture_mismatches': int((got.astype(np.uint8) != want).sum()), 'random_mismatches': int((a != b).sum()),
            'thin_mismatches': int((c != d).sum()), 'thin_painted': int((c != 0).sum()),
            'painted': int((a != 0).sum()),
            'offset_mismatches': int((offset != torch.from_numpy(g['instance/offset'])).sum()),
            'flow_mismatches': int((flow != torch.from_numpy(g['instance/flow'])).sum()),
            'center_err': err(center, g['instance/center'])}


def image_prep(ops):
    """csrc/stp3_image.hip through stp3_amd.datas.ImagePreprocessor (GPU route) against the torch statements of the same
    module (CPU route; byte-exact with Pillow, tests/test_datas_cpu.py): float32 and bf16 output, a crop inside the
    resized image and one that is padded with zeros, up- and down-scaling (<= 9 taps: the dword path; 11 taps: the generic
    one), a source row length that is not a multiple of 16 bytes."""
    from stp3_amd.datas import ImagePreprocessor
    from tests import helpers as H
    is_cuda = torch.Tensor.is_cuda
    out = {}
    cases = {'down': ((3, 90, 160, 3), (48, 27), (2, 5, 46, 25)), 'padded': ((3, 90, 160, 3), (48, 27), (-3, 5, 51, 30)),
             'up_odd': ((2, 37, 53, 3), (70, 50), (3, 2, 64, 45)), 'mixed': ((1, 64, 21, 3), (40, 20), (0, 0, 40, 20)),
             'down5_generic_taps': ((1, 50, 200, 3), (40, 10), (1, 0, 39, 10))}
    for name, (shape, resize_dims, crop) in cases.items():
        images = torch.from_numpy(H.image_bytes(shape, 420 + len(name)))
        prep = ImagePreprocessor(resize_dims=resize_dims, crop=crop, source_hw=shape[1:3])
        torch.Tensor.is_cuda = is_cuda
        y32, y16 = prep(images), prep(images, out_dtype=torch.bfloat16)
        torch.Tensor.is_cuda = property(lambda self: False)
        ref = prep(images)
        torch.Tensor.is_cuda = is_cuda
        out[name] = {'f32_mismatches': int((y32 != ref).sum()), 'bf16_mismatches': int((y16 != ref.to(torch.bfloat16)).sum())}
    return out


def aspp_join(ops):
    """ASPP (stp3/layers/convolutions.py:217-270) in training mode on bf16 activations: the spatial branches write their
    results into the channel slices of ONE buffer (ops_fused.join_slices: no concatenation copy), the dilated branches run
    through the fused conv -> BatchNorm operator, the branches' input gradients are added in one pass (ops.fan_out) --
    against the module's own torch statements in float32 on the same bf16-representable data (CPU route)."""
    import torch.nn as nn
    from stp3_amd.layers.convolutions import ASPP
    from stp3_amd import ops_fused
    is_cuda = torch.Tensor.is_cuda
    out = {}
    torch.manual_seed(31)
    for name, (cin, co, rates, hw) in {'all_fused': (16, 16, (1, 2, 3), (9, 10)), 'one_sliced': (16, 8, (2, 3, 12), (9, 10))}.items():
        m = ASPP(cin, rates, co).train()
        m.project[3].p = 0.0
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(p.to(torch.bfloat16).float())
        x0 = torch.randn(2, cin, *hw).to(torch.bfloat16)
        gy = None
        res, joined = [], []
        real_join = ops_fused.join_slices
        ops_fused.join_slices = lambda buf, parts: (joined.append(len(parts)), real_join(buf, parts))[1]
        for kernel in (True, False):
            torch.Tensor.is_cuda = is_cuda if kernel else property(lambda self: False)
            m.zero_grad()
            x = (x0.clone().contiguous(memory_format=torch.channels_last) if kernel else x0.float()).requires_grad_()
            with torch.autocast('cpu', dtype=torch.bfloat16, enabled=kernel):
                y = m(x)
            if gy is None:
                gy = torch.randn(y.shape).to(torch.bfloat16)
            y.backward(gy.to(y.dtype))
            res.append((y.detach().float(), x.grad.float(), torch.cat([p.grad.flatten() for p in m.parameters()])))
        torch.Tensor.is_cuda = is_cuda
        ops_fused.join_slices = real_join
        out[name] = {'y': rel(res[0][0], res[1][0]), 'dx': rel(res[0][1], res[1][1]), 'dparam': rel(res[0][2], res[1][2]),
                     'joined': joined}
    return out


def pointwise_bn(ops):
    """ops_fused._PointwiseBnAct (1x1 convolution -> BatchNorm -> activation, the convolution output recomputed instead of
    stored: stp3_conv2d_fwd_stats / _fwd_bnact / _bn_bwd_reduce / _bn_bwd_apply) against ops_fused._ConvBnAct (the stored
    route, same kernels otherwise): outputs, all gradients and the running statistics -- the two routes round at the same
    places, so they agree to float32 summation order."""
    import torch.nn as nn
    from stp3_amd import ops_fused
    out = {}
    torch.manual_seed(51)
    for name, (n, cin, cout, h, w, act) in {'swish_24_144': (3, 24, 144, 9, 11, ops.ACT_SWISH), 'relu_160_960': (2, 160, 136, 5, 7, ops.ACT_RELU),
                                            'none_8_48': (2, 8, 48, 13, 10, ops.ACT_NONE), 'swish_ragged': (1, 32, 192, 11, 13, ops.ACT_SWISH)}.items():
        conv = nn.Conv2d(cin, cout, 1, bias=False)
        res = []
        x0 = torch.randn(n, cin, h, w).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(n, cout, h, w).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        state = None
        for recompute in (True, False):
            bn = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).train()
            if state is None:
                with torch.no_grad():
                    bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
                state = {k: v.clone() for k, v in bn.state_dict().items()}
            bn.load_state_dict(state)
            conv.zero_grad()
            x = x0.clone().requires_grad_()
            assert ops_fused.pointwise_bn_act_supported(x, conv, bn)
            if recompute:
                y = ops_fused.pointwise_bn_act(x, conv, bn, act, group=False)
            else:
                y = ops_fused.conv_bn_act(x, conv.weight, None, bn, act, group=False)
            y.backward(gy)
            res.append([y.detach().float(), x.grad.float(), conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                        bn.running_mean.clone(), bn.running_var.clone()])
        out[name] = {k: rel(a, b) for k, a, b in zip(('y', 'dx', 'dw', 'dgamma', 'dbeta', 'rmean', 'rvar'), res[0], res[1])}
    # the data gradient out of the apply pass (stp3_conv2d_bn_bwd_apply_dx) against the separate data-gradient convolution,
    # without and with a skip gradient handed over by a carrier (ops.SkipCarrier); ragged pixel counts, both widths
    from stp3_amd import _lib
    lib = _lib.lib()
    for name, (n, cin, cout, h, w) in {'dx_24_144': (3, 24, 144, 9, 11), 'dx_32_192': (1, 32, 192, 11, 13), 'dx_8_192': (2, 8, 192, 5, 7)}.items():
        conv = nn.Conv2d(cin, cout, 1, bias=False)
        bn = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).train()
        x0 = torch.randn(n, cin, h, w).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(n, cout, h, w).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        skip = torch.randn(n, cin, h, w).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        worst, ran = 0.0, []
        for with_skip in (False, True):
            got = []
            for on in (False, True):
                ops_fused.EXPAND_DGRAD_IN_APPLY = on
                conv.zero_grad(); bn.zero_grad()
                carrier = None
                if with_skip:
                    carrier = ops.SkipCarrier()
                    carrier.grad = skip.clone()
                x = x0.clone().requires_grad_()
                y = ops_fused.pointwise_bn_act(x, conv, bn, ops.ACT_SWISH, group=False, skip_carrier=carrier)
                y.backward(gy)
                got.append([x.grad.float(), conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()])
                ran.append(carrier is None or carrier.grad is None)
            worst = max(worst, max(rel(a, b) for a, b in zip(got[1], got[0])))
        ops_fused.EXPAND_DGRAD_IN_APPLY = True
        out[name] = {'one_vs_two_kernels': worst, 'carrier_emptied': all(ran)}
    return out


def pointwise_stream(ops):
    """The streaming kernels for short-contraction 1x1 convolutions (pointwise_rows_kernel / pointwise_direct_kernel in
    stp3_conv.hip: every 1x1 / stride-1 layer with <= 128 input and >= 64 output channels) in their five modes: the stored route
    (conv + statistics -> BatchNorm -> act, data gradient of the NEXT layer's shape) and the recomputing route against
    float32 torch on the same bf16-representable data.  Shapes: ragged pixel counts (last 32-pixel tile partial), Cin that is
    no multiple of 16 (zero k tail), both channel-tile widths with a ragged last tile."""
    import torch.nn as nn
    import torch.nn.functional as F
    from stp3_amd import ops_fused
    from stp3_amd.layers import fused
    out = {}
    cl = torch.channels_last
    for name, (n, cin, cout, h, w, act) in {'24_144': (3, 24, 144, 9, 11, ops.ACT_SWISH), '32_192': (1, 32, 192, 11, 13, ops.ACT_SWISH),
                                            '56_336': (2, 56, 336, 5, 9, ops.ACT_RELU), '112_672': (1, 112, 672, 6, 7, ops.ACT_SWISH),
                                            '128_64': (2, 128, 64, 7, 5, ops.ACT_NONE), '8_72': (5, 8, 72, 16, 17, ops.ACT_SWISH),
                                            # whole-line tiles of 128 channels (the kernel of the outputs beyond the infinity cache)
                                            '72_256': (2, 72, 256, 7, 9, ops.ACT_SWISH), '40_128': (1, 40, 128, 9, 5, ops.ACT_RELU)}.items():
        g = torch.Generator().manual_seed(11)
        x0 = torch.randn(n, cin, h, w, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
        gy = torch.randn(n, cout, h, w, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
        w0 = (torch.randn(cout, cin, 1, 1, generator=g) * 0.2).to(torch.bfloat16).float()
        res = []
        for mode in ('stored', 'recompute', 'torch'):
            conv = nn.Conv2d(cin, cout, 1, bias=False)
            with torch.no_grad():
                conv.weight.copy_(w0)
            bn = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).train()
            with torch.no_grad():
                bn.weight.copy_(torch.linspace(0.5, 1.5, cout)); bn.bias.copy_(torch.linspace(-0.3, 0.3, cout))
            x = (x0.float() if mode == 'torch' else x0.clone()).requires_grad_()
            if mode == 'stored':
                y = ops_fused.conv_bn_act(x, conv.weight, None, bn, act, group=False)
            elif mode == 'recompute':
                y = ops_fused.pointwise_bn_act(x, conv, bn, act, group=False)
            else:
                y = fused.bn_act_reference(bn, F.conv2d(x, conv.weight), act, None, ops.RES_NONE, None, None)
            y.backward(gy.float() if mode == 'torch' else gy)
            res.append([y.detach().float(), x.grad.float(), conv.weight.grad, bn.weight.grad, bn.bias.grad, bn.running_mean,
                        bn.running_var])
        # plain mode as a data gradient (K = Cout of the layer, N = Cin): dX = dY * W, float32 torch beside it
        kk = min(cout, 128)
        xg = gy[:, :kk].contiguous(memory_format=cl)
        wg = (torch.randn(96, kk, 1, 1, generator=g) * 0.2).to(torch.bfloat16).float()
        dx = ops.conv2d(xg, wg, None, 1, 0, 1)
        dxr = F.conv2d(xg.float(), wg)
        out[name] = {'stored_vs_torch': max(rel(a, b) for a, b in zip(res[0], res[2])),
                     'recompute_vs_stored': max(rel(a, b) for a, b in zip(res[1], res[0])),
                     'plain_vs_torch': rel(dx.float(), dxr)}
    return out


def decoder_heads(ops):
    """Decoder (stp3/models/decoder.py:8-140) in training mode on bf16 activations with the heads that read the same
    tensor MERGED (one 3x3 convolution + one BatchNorm over all heads' channels, one block-diagonal 1x1 convolution)
    against the same kernels run head by head: same outputs, gradients, running statistics and batch counters."""
    from stp3_amd.models import decoder as D
    from stp3_amd.layers import fused
    gate = {'perceive_hdmap': True, 'predict_pedestrian': True, 'predict_instance': True, 'predict_future_flow': True,
            'planning': False}
    torch.manual_seed(41)
    # the stand-in's way of switching the GPU autocast on (as the whole-step cases do)
    torch.is_autocast_enabled = lambda *a: True
    torch.get_autocast_gpu_dtype = lambda: torch.bfloat16
    torch.get_autocast_dtype = lambda *a: torch.bfloat16
    m = D.Decoder(16, 2, 3, 2, gate).train()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    x0 = torch.randn(1, 3, 16, 16, 16)
    res, gouts = [], None
    for merge in (True, False):
        D.MERGE_HEADS = merge
        m.load_state_dict(state)
        m.zero_grad()
        x = x0.clone().requires_grad_()
        with torch.autocast('cpu', dtype=torch.bfloat16):
            o = m(x)
        keys = [k for k, v in o.items() if v is not None]
        if gouts is None:
            gouts = {k: torch.randn(o[k].shape) for k in keys}
        sum((o[k].float() * gouts[k]).sum() for k in keys).backward()
        fused.flush_batch_counters()
        heads = {n: p.grad.clone() for n, p in m.named_parameters() if '_head' in n}
        rest = {n: p.grad.clone() for n, p in m.named_parameters() if '_head' not in n}
        stats = {n: b.clone() for n, b in m.named_buffers() if '_head' in n}
        res.append((torch.cat([o[k].float().flatten() for k in keys]), x.grad.clone(), heads, rest, stats))
    D.MERGE_HEADS = True
    a, b = res
    out = {'y': rel(a[0], b[0]), 'dx': rel(a[1], b[1]),
           'head_grads': max(rel(a[2][n], b[2][n]) for n in a[2]),
           # (relative L2 over all of them: biases in front of a BatchNorm have noise-level gradients of their own)
           'other_grads': float((torch.cat([(a[3][n] - b[3][n]).flatten() for n in a[3]]).norm()
                                 / torch.cat([b[3][n].flatten() for n in a[3]]).norm())),
           'running_stats': max(rel(a[4][n].double(), b[4][n].double()) for n in a[4]), 'n_head_params': len(a[2]),
           'aliased': bool(m.segmentation_head[1].running_mean.data_ptr() + 4 * 16 == m.pedestrian_head[1].running_mean.data_ptr())}
    # the state dict still holds one entry per head with the right values
    sd = m.state_dict()
    out['state_dict_ok'] = bool(all(k in sd for k in state) and sd['pedestrian_head.1.running_mean'].shape == (16,))
    return out


def conv_bn(ops):
    """conv -> BatchNorm -> act (+ skip / drop-connect) as ONE operator (conv v2 with the statistics in its epilogue)
    against the two separate operators, and both against float32 torch on the same bf16-representable data."""
    import torch.nn as nn
    import torch.nn.functional as F
    from stp3_amd import ops_fused
    from stp3_amd.layers import fused
    out = {}
    cases = {'3x3 relu': (16, 24, 3, 1, ops.ACT_RELU, ops.RES_NONE, False),
             '1x1 project + drop-connect + skip': (40, 16, 1, 0, ops.ACT_NONE, ops.RES_AFTER_ACT, True),
             '3x3 swish + skip before': (16, 16, 3, 1, ops.ACT_SWISH, ops.RES_BEFORE_ACT, False)}
    for name, (cin, cout, k, pad, act, rm, with_oscale) in cases.items():
        g = torch.Generator().manual_seed(7)
        cl = torch.channels_last
        x0 = torch.randn(4, cin, 7, 10, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
        r0 = torch.randn(4, cout, 7, 10, generator=g).to(torch.bfloat16).contiguous(memory_format=cl) if rm else None
        gy = torch.randn(4, cout, 7, 10, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
        osc = torch.tensor([1.25, 0.0, 1.25, 1.25]) if with_oscale else None
        w0 = (torch.randn(cout, cin, k, k, generator=g) * 0.2).to(torch.bfloat16).float()
        res = []
        for mode in ('fused', 'separate', 'torch'):
            conv = nn.Conv2d(cin, cout, k, padding=pad, bias=False)
            with torch.no_grad():
                conv.weight.copy_(w0)
            bn = nn.BatchNorm2d(cout)
            x = (x0.float() if mode == 'torch' else x0.clone()).requires_grad_()
            r = None if r0 is None else (r0.float() if mode == 'torch' else r0.clone()).requires_grad_()
            if mode == 'fused':
                y = ops_fused.conv_bn_act(x, conv.weight, None, bn, act, r, rm, 1, pad, 1, group=False, oscale=osc)
            elif mode == 'separate':
                y = fused.bn_act(bn, ops.conv2d(x, conv.weight, None, 1, pad, 1), act, res=r, res_mode=rm, oscale=osc)
            else:
                y = fused.bn_act_reference(bn, F.conv2d(x, conv.weight, None, 1, pad), act, r, rm, None, osc)
            y.backward(gy.float() if mode == 'torch' else gy)
            res.append([y.detach().float(), x.grad.float(), conv.weight.grad, bn.weight.grad, bn.bias.grad, bn.running_mean,
                        bn.running_var] + ([r.grad.float()] if r is not None else []))
        out[name] = {'fused_vs_separate': max(rel(a, b) for a, b in zip(res[0], res[1])),
                     'fused_vs_torch_f32': max(rel(a, b) for a, b in zip(res[0], res[2]))}
    return out


def _model_step(ops, autocast, flags=None, bn_eval=False, full_losses=False):
    """One whole training step (encoder, lift, temporal model, decoder, losses, backward) through the kernels, against
    the CPU port of the same model (oracle/cpu_model.py: reference-algorithm lift, plain torch everywhere else)."""
    import torch.nn as nn
    from oracle.cpu_model import CpuPortSTP3
    from stp3_amd import synthetic
    from stp3_amd.config import perception_cfg
    from stp3_amd.trainer import TrainingModule
    from stp3_amd.utils import to_channels_last
    for mod, name, val in (flags or []):
        setattr(mod, name, val)
    extra = {'LIFT.GT_DEPTH': True, 'INSTANCE_SEG.ENABLED': True, 'INSTANCE_FLOW.ENABLED': True} if full_losses else {}
    cfg = perception_cfg(**{'IMAGE.FINAL_DIM': (64, 96), 'LIFT.X_BOUND': [-10.0, 10.0, 0.5], 'LIFT.Y_BOUND': [-10.0, 10.0, 0.5],
                            'LIFT.D_BOUND': [2.0, 10.0, 1.0], **extra})
    g = torch.Generator().manual_seed(3)
    intr, extr, ego = synthetic.make_rig(1, 3, 6, (64, 96), seed=3)
    batch = {'image': torch.randn(1, 3, 6, 3, 64, 96, generator=g), 'intrinsics': intr, 'extrinsics': extr,
             'future_egomotion': ego, 'segmentation': (torch.rand(1, 3, 1, 40, 40, generator=g) > 0.9).long(),
             'pedestrian': (torch.rand(1, 3, 1, 40, 40, generator=g) > 0.95).long(),
             'hdmap': (torch.rand(1, 3, 2, 40, 40, generator=g) > 0.7).long(), 'gt_trajectory': torch.zeros(1, 3, 3)}
    if full_losses:          # bench.py's default workload: depth cross-entropy + instance centerness / offset + flow
        batch['instance'], batch['centerness'], batch['offset'], batch['flow'] = \
            synthetic.make_instance_labels(batch['segmentation'], seed=3)
        batch['depths'] = torch.randint(0, 14, (1, 3, 6, 64, 96), generator=g).float()

    def quiet(module):
        module.train()
        for m in module.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        module.model.encoder.backbone._global_params.drop_connect_rate = 0.0
        if bn_eval:            # running statistics: without batch statistics over 4x6 maps the step is a smooth function
            for m in module.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return module

    torch.manual_seed(11)
    module = quiet(to_channels_last(TrainingModule(cfg.convert_to_dict())))
    state = {k: v.clone() for k, v in module.model.state_dict().items()}
    # ---- reference first, with tensors that say what they are (CPU): plain torch + the oracle's lift ----
    del torch.Tensor.is_cuda
    ref = quiet(TrainingModule(cfg.convert_to_dict()))
    port = CpuPortSTP3(cfg)
    port.load_state_dict(state, strict=False)
    for name in ('segmentation_weight', 'pedestrian_weight', 'hdmap_weight', 'depths_weight', 'centerness_weight',
                 'offset_weight', 'flow_weight'):
        if hasattr(ref.model, name):
            setattr(port, name, getattr(ref.model, name))
    ref.model = port
    quiet(ref)
    ref_loss = ref.training_step(batch)
    ref_loss.backward()
    ref_grads = {n: p.grad.clone() for n, p in ref.model.named_parameters() if p.grad is not None}
    # ---- the GPU code path on the kernels ----
    torch.Tensor.is_cuda = property(lambda self: True)
    if autocast:
        torch.is_autocast_enabled = lambda *a: True
        torch.get_autocast_gpu_dtype = lambda: torch.bfloat16
        torch.get_autocast_dtype = lambda *a: torch.bfloat16
    module.model.prepare_plan(intr, extr, ego, torch.device('cpu'))
    if autocast:
        with torch.autocast('cpu', dtype=torch.bfloat16):
            loss = module.training_step(batch)
    else:
        loss = module.training_step(batch)
    loss.backward()
    num = den = 0.0
    worst, worst_name, missing = 0.0, '', []
    groups = {}

    def group_of(n):                                  # from the loss backwards: decoder, temporal model, encoder heads, trunk
        parts = n.split('.')
        if parts[0] != 'encoder':
            return parts[0]
        if parts[1] != 'backbone':
            return 'encoder.' + parts[1]
        return 'trunk.' + (f'block{int(parts[3]):02d}' if parts[2] == '_blocks' else parts[2])

    for n, p in module.model.named_parameters():
        if n not in ref_grads:
            continue
        if p.grad is None:
            missing.append(n)
            continue
        a, b = p.grad.double(), ref_grads[n].double()
        num += float((a - b).pow(2).sum())
        den += float(b.pow(2).sum())
        gacc = groups.setdefault(group_of(n), [0.0, 0.0])
        gacc[0] += float((a - b).pow(2).sum())
        gacc[1] += float(b.pow(2).sum())
        e = float((a - b).norm() / b.norm().clamp_min(1e-12))
        if e > worst and float(b.norm()) > 1e-6:
            worst, worst_name = e, n
    return {'loss': float(loss), 'ref_loss': float(ref_loss), 'grad_rel_l2': (num / max(den, 1e-30)) ** 0.5,
            'worst_param_rel_l2': worst, 'worst_param': worst_name, 'params_without_grad': missing,
            'grad_rel_l2_by_group': {k: round((v[0] / max(v[1], 1e-30)) ** 0.5, 4) for k, v in sorted(groups.items())}}


def _kernel_path_grads(lib_path, samples, rank=None, world=1, port=0, out=None, local_stats=False):
    """Float32 training step of the GPU code path on the kernels for the given samples of a fixed 2-sample batch;
    with world > 1 the process joins a gloo group first (cross-replica BatchNorm statistics, bucketed gradients)."""
    import torch.distributed as dist
    import torch.nn as nn
    ops = setup(lib_path)
    from stp3_amd import synthetic
    from stp3_amd.config import perception_cfg
    from stp3_amd.parallel import GradientBuckets
    from stp3_amd.trainer import TrainingModule
    from stp3_amd.utils import to_channels_last
    torch.set_num_threads(2)
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group('gloo', rank=rank, world_size=world)
    cfg = perception_cfg(**{'IMAGE.FINAL_DIM': (64, 96), 'LIFT.X_BOUND': [-10.0, 10.0, 0.5], 'LIFT.Y_BOUND': [-10.0, 10.0, 0.5],
                            'LIFT.D_BOUND': [2.0, 10.0, 1.0], 'SEMANTIC_SEG.VEHICLE.USE_TOP_K': False,
                            'SEMANTIC_SEG.PEDESTRIAN.USE_TOP_K': False, 'SEMANTIC_SEG.HDMAP.USE_TOP_K': [False, False]})
    g = torch.Generator().manual_seed(3)
    intr, extr, ego = synthetic.make_rig(2, 3, 6, (64, 96), seed=3)
    full = {'image': torch.randn(2, 3, 6, 3, 64, 96, generator=g), 'intrinsics': intr, 'extrinsics': extr, 'future_egomotion': ego,
            'segmentation': (torch.rand(2, 3, 1, 40, 40, generator=g) > 0.9).long(),
            'pedestrian': (torch.rand(2, 3, 1, 40, 40, generator=g) > 0.95).long(),
            'hdmap': (torch.rand(2, 3, 2, 40, 40, generator=g) > 0.7).long(), 'gt_trajectory': torch.zeros(2, 3, 3)}
    batch = {k: v[samples] for k, v in full.items()}
    torch.manual_seed(11)
    module = to_channels_last(TrainingModule(cfg.convert_to_dict())).train()
    for m in module.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if local_stats and isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.stp3_local_stats = True                                   # negative control: per-rank statistics
    module.model.encoder.backbone._global_params.drop_connect_rate = 0.0
    buckets = GradientBuckets(module.model)
    module.model.prepare_plan(batch['intrinsics'], batch['extrinsics'], batch['future_egomotion'], torch.device('cpu'))
    buckets.zero_grad()
    loss = module.training_step(batch)
    loss.backward()
    buckets.finish()
    flat = torch.cat([f.clone() for f, _ in buckets.buckets])
    if world > 1:
        out[rank] = (float(loss), flat)
        dist.destroy_process_group()
        return None
    return float(loss), flat


def _two_rank_worker(rank, lib_path, port, out, local_stats):
    _kernel_path_grads(lib_path, slice(rank, rank + 1), rank=rank, world=2, port=port, out=out, local_stats=local_stats)


def model_step_two_ranks(ops):
    """2 gloo ranks x 1 sample through the kernels (split BatchNorm operator: statistics | all-reduce | apply; bucketed
    gradient all-reduce) against 1 process x 2 samples through the same kernels (the composite BatchNorm operator)."""
    import socket
    import torch.multiprocessing as mp
    lib_path = sys.argv[1]

    def two(local_stats):
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        mgr = mp.Manager()
        res = mgr.dict()
        mp.spawn(_two_rank_worker, args=(lib_path, port, res, local_stats), nprocs=2, join=True)
        return {k: v for k, v in res.items()}

    out = two(False)
    loss, ref = _kernel_path_grads(lib_path, slice(0, 2))
    ctl = two(True)

    def rel_l2(a):
        return float((a.double() - ref.double()).norm() / ref.double().norm())
    return {'ranks_identical': bool(torch.equal(out[0][1], out[1][1])), 'loss_two_rank_mean': 0.5 * (out[0][0] + out[1][0]),
            'loss_one_process': loss, 'grad_rel_l2': rel_l2(out[0][1]), 'grad_rel_l2_per_rank_statistics': rel_l2(ctl[0][1])}


def _group_worker(rank, lib_path, port, out):
    """One rank of ``bn_group_two_ranks``: three sibling layers of one input -- a fused conv -> BatchNorm -> ReLU (bf16,
    statistics from the convolution epilogue), a BatchNorm + ReLU with a per-sample bias on zero-padded 40-lane rows, a
    plain BatchNorm on float32 -- first as three operators (three exchanges per pass), then as ONE exchange group."""
    import torch.distributed as dist
    import torch.nn as nn
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2')
    dist.init_process_group('gloo', rank=rank, world_size=2)
    setup(lib_path)
    from stp3_amd.layers import fused
    torch.manual_seed(5)
    cl = torch.channels_last
    conv = nn.Conv2d(16, 24, 3, padding=1, bias=False)
    bns = [nn.BatchNorm2d(24), nn.BatchNorm2d(35), nn.BatchNorm2d(8)]
    for bn in bns:
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.2)
    torch.manual_seed(100 + rank)                      # every rank its own shard
    x0 = torch.randn(2, 16, 6, 7).to(torch.bfloat16).contiguous(memory_format=cl)
    y1 = torch.zeros(2, 40, 6, 7).contiguous(memory_format=cl)
    y1[:, :35] = torch.randn(2, 35, 6, 7)
    y2 = torch.randn(2, 8, 6, 7).contiguous(memory_format=cl)
    sb = torch.randn(2, 35) * 0.3
    gys = [torch.randn(2, 24, 6, 7).to(torch.bfloat16).contiguous(memory_format=cl), torch.randn(2, 40, 6, 7).contiguous(memory_format=cl),
           torch.randn(2, 8, 6, 7).contiguous(memory_format=cl)]
    calls = {'n': 0}
    real = dist.all_reduce

    def counting(*a, **k):
        calls['n'] += 1
        return real(*a, **k)
    dist.all_reduce = counting
    results = []
    import copy
    for grouped in (False, True):
        cv, b0, b1, b2 = copy.deepcopy(conv), *[copy.deepcopy(b) for b in bns]
        x, a1, a2, s1 = (t.clone().requires_grad_() for t in (x0, y1, y2, sb))
        calls['n'] = 0
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=False):
            members = [fused.conv_bn_act_member(x, cv, b0, fused.ACT_RELU), dict(bn=b1, x=a1, act=fused.ACT_RELU, sbias=s1),
                       dict(bn=b2, x=a2)]
            assert not isinstance(members[0], dict), 'the convolution member must take the fused operator'
            outs = fused.bn_act_group(members) if grouped else [fused._run_member(m) for m in members]
        fwd_calls = calls['n']
        torch.autograd.backward(outs, gys)
        grads = [x.grad, cv.weight.grad, b0.weight.grad, b0.bias.grad, a1.grad, s1.grad, b1.weight.grad, b1.bias.grad, a2.grad,
                 b2.weight.grad, b2.bias.grad, b0.running_var, b1.running_mean, b2.running_var]
        results.append((fwd_calls, calls['n'] - fwd_calls, [o.detach().float() for o in outs], [g.detach().float().clone() for g in grads]))
    dist.all_reduce = real
    (f0, b0c, o0, g0), (f1, b1c, o1, g1) = results
    out[rank] = {'exchanges_separate': (f0, b0c), 'exchanges_grouped': (f1, b1c),
                 'outputs_equal': all(torch.equal(a, b) for a, b in zip(o0, o1)),
                 'grads_equal': all(torch.equal(a, b) for a, b in zip(g0, g1))}
    dist.destroy_process_group()


def bn_group_two_ranks(ops):
    """ops_fused._ExchangeGroup on two gloo ranks: the same bits as the separate operators, one exchange per pass."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    res = mgr.dict()
    mp.spawn(_group_worker, args=(sys.argv[1], port, res), nprocs=2, join=True)
    return {f'rank{k}': v for k, v in res.items()}


def model_step_f32(ops):            # float32: BatchNorm / depthwise / voxel-pool kernels (dense convolutions stay on torch)
    return _model_step(ops, autocast=False)


def model_step_f32_full_losses(ops):    # BASELINE configs[2] = bench.py's default: + depth CE, instance centerness / offset, flow
    return _model_step(ops, autocast=False, full_losses=True)


def model_step_bf16(ops):           # bf16 autocast: the MFMA convolution kernels too
    return _model_step(ops, autocast=True)


def model_step_bf16_bn_eval(ops):   # the same with BatchNorm on its running statistics (see _model_step)
    return _model_step(ops, autocast=True, bn_eval=True)


def model_step_f32_bn_eval(ops):    # float32 with BatchNorm on its running statistics: a smooth function, kernels to round-off
    return _model_step(ops, autocast=False, bn_eval=True)


def fuzz(ops, seed=1):
    """Random shapes / modes of the depthwise, BatchNorm and dense convolution operators against torch in float32."""
    import random
    import torch.nn.functional as F
    from stp3_amd.layers import fused
    cl = torch.channels_last
    bad = []
    random.seed(seed)
    def chk(tag, cfgd, pairs, tol):
        for name, a, b in pairs:
            e = rel(a, b)
            if not (e <= tol):
                bad.append((tag, cfgd, name, e))
    # ---- depthwise
    for it in range(40):
        c = random.choice([8, 24, 40, 64]); k = random.choice([3, 5]); s = random.choice([1, 2])
        h, w = random.randint(k, 11), random.randint(k, 13); n = random.randint(1, 3)
        pad = tuple(random.randint(0, k // 2 + 1) for _ in range(4))
        dt = random.choice([torch.float32, torch.bfloat16])
        cfgd = dict(c=c, k=k, s=s, h=h, w=w, n=n, pad=pad, dt=str(dt))
        try:
            x0 = torch.randn(n, c, h, w).to(dt).contiguous(memory_format=cl); w0 = torch.randn(c, 1, k, k) * 0.3
            if (h + pad[2] + pad[3] - k) < 0 or (w + pad[0] + pad[1] - k) < 0: continue
            x, wt = x0.clone().requires_grad_(), w0.clone().requires_grad_()
            y = ops.depthwise_conv2d(x, wt, s, pad); gy = torch.randn_like(y); y.backward(gy)
            xr, wr = x0.float().requires_grad_(), w0.clone().requires_grad_()
            yr = F.conv2d(F.pad(xr, pad), wr, None, s, 0, 1, c); yr.backward(gy.float())
            tol = 1e-4 if dt == torch.float32 else 3e-2
            chk('dw', cfgd, [('y', y.detach().float(), yr.detach()), ('dx', x.grad.float(), xr.grad), ('dw', wt.grad, wr.grad)], tol)
        except Exception as e:
            bad.append(('dw', cfgd, 'EXC', repr(e)))
    # ---- batchnorm
    for it in range(60):
        n = random.randint(2, 4); c = random.choice([1, 3, 7, 8, 9, 16, 24, 33, 40, 64, 70]); h, w = random.randint(1, 9), random.randint(2, 9)
        if it % 6 == 5: h = w = 1                                # 1x1 descriptors (ASPP / pyramid pooling) run on the kernels too
        dt = random.choice([torch.float32, torch.bfloat16]); training = random.random() < 0.7
        act = random.choice([ops.ACT_NONE, ops.ACT_RELU, ops.ACT_SWISH]); rm = random.choice([ops.RES_NONE, ops.RES_BEFORE_ACT, ops.RES_AFTER_ACT])
        with_sb, with_os = random.random() < 0.3, random.random() < 0.3
        sliced = random.random() < 0.3
        padded = (not sliced) and c % 8 != 0 and random.random() < 0.5     # zero-padded channel lanes (stp3_bn_dims.cpad)
        cp = (c + 7) // 8 * 8
        cfgd = dict(n=n, c=c, h=h, w=w, dt=str(dt), tr=training, act=act, rm=rm, sb=with_sb, os=with_os, sliced=sliced,
                    padded=padded)
        try:
            bn_a, bn_b = torch.nn.BatchNorm2d(c, eps=1e-3), torch.nn.BatchNorm2d(c, eps=1e-3)
            with torch.no_grad():
                bn_a.weight.uniform_(0.5, 1.5); bn_a.bias.normal_(0, 0.2); bn_a.running_mean.normal_(0, 0.2); bn_a.running_var.uniform_(0.5, 1.5)
            bn_b.load_state_dict(bn_a.state_dict()); bn_a.train(training); bn_b.train(training)
            if sliced:
                wide = torch.randn(n, c + 8, h, w).to(dt).contiguous(memory_format=cl); x0 = wide[:, 3:3 + c]
            else:
                x0 = torch.randn(n, c, h, w).to(dt).contiguous(memory_format=cl)
            r0 = torch.randn(n, c, h, w).to(dt).contiguous(memory_format=cl) if rm else None
            sb0 = torch.randn(n, c) * 0.3 if with_sb else None; osc = torch.rand(n) + 0.5 if with_os else None
            gy = torch.randn(n, c, h, w).to(dt).contiguous(memory_format=cl)
            res = []
            def lanes(t):                  # the tensor in cp-lane rows, NaN in the padding lanes
                full = torch.full((n, cp, h, w), float('nan'), dtype=t.dtype).contiguous(memory_format=cl)
                full[:, :c] = t
                return full
            for kernel, bn in ((True, bn_a), (False, bn_b)):
                cast = (lambda t: t.clone()) if kernel else (lambda t: t.float())
                pad_in = lanes if (kernel and padded) else (lambda t: t)
                x = pad_in(cast(x0).detach()).requires_grad_(); r = pad_in(cast(r0)).requires_grad_() if r0 is not None else None
                sb = sb0.clone().requires_grad_() if sb0 is not None else None
                y = (fused.bn_act if kernel else fused.bn_act_reference)(bn, x, act, r, rm, sb, osc)
                y.backward(pad_in(gy) if kernel else gy.float())
                cut = (lambda t: t[:, :c]) if (kernel and padded) else (lambda t: t)
                if kernel and padded and not (bool((y.detach()[:, c:] == 0).all()) and bool((x.grad[:, c:] == 0).all())):
                    bad.append(('bn', cfgd, 'padding lanes not zero', 0.0))
                res.append([('y', cut(y.detach()).float()), ('dx', cut(x.grad).float()), ('dg', bn.weight.grad), ('db', bn.bias.grad)] + ([('dr', cut(r.grad).float())] if r is not None else []) + ([('dsb', sb.grad)] if sb is not None else []) + [('rm', bn.running_mean.clone()), ('rv', bn.running_var.clone())])
            tol = 5e-4 if dt == torch.float32 else 3e-2          # float32: sums of <= 40 values that cancel (C = 1, 3 x 4 maps)
            chk('bn', cfgd, [(a[0], a[1], b[1]) for a, b in zip(*res)], tol)
        except Exception as e:
            bad.append(('bn', cfgd, 'EXC', repr(e)))
    # ---- dense conv
    for it in range(40):
        cin = random.choice([8, 16, 24, 40, 72]); cout = random.choice([1, 2, 3, 8, 24, 35, 70]); k = random.choice([1, 3, 5, 7]); s = random.choice([1, 2])
        d = 1 if k == 1 else random.choice([1, 1, 2, 3])        # torch's own CPU convolution corrupts the heap for 1x1 + dilation
        p = random.choice([0, (k - 1) // 2 * d, d]); n = random.randint(1, 2)
        lo = max(1, d * (k - 1) + 1 - 2 * p)
        h, w = random.randint(lo, max(lo, 12)), random.randint(lo, max(lo, 14))
        if h + 2 * p - d * (k - 1) < 1 or w + 2 * p - d * (k - 1) < 1: continue
        bias = random.random() < 0.5
        cfgd = dict(cin=cin, cout=cout, k=k, s=s, d=d, p=p, n=n, h=h, w=w, bias=bias)
        try:
            x0 = torch.randn(n, cin, h, w).to(torch.bfloat16).contiguous(memory_format=cl)
            w0 = (torch.randn(cout, cin, k, k) * 0.2).to(torch.bfloat16).float(); b0 = torch.randn(cout) if bias else None
            x, wt = x0.clone().requires_grad_(), w0.clone().requires_grad_(); b = b0.clone().requires_grad_() if bias else None
            y = ops.conv2d(x, wt, b, s, p, d, out_dtype=torch.float32); gy = torch.randn_like(y).to(torch.bfloat16).float(); y.backward(gy)
            xr, wr = x0.float().requires_grad_(), w0.clone().requires_grad_(); br = b0.clone().requires_grad_() if bias else None
            yr = F.conv2d(xr, wr, br, s, p, d); yr.backward(gy)
            pairs = [('y', y.detach(), yr.detach()), ('dx', x.grad.float(), xr.grad), ('dw', wt.grad, wr.grad)] + ([('db', b.grad, br.grad)] if bias else [])
            chk('conv', cfgd, pairs, 2e-2)
        except Exception as e:
            bad.append(('conv', cfgd, 'EXC', repr(e)))
    return {'problems': [[t, str(c), n, str(e)] for t, c, n, e in bad]}


def layernorm(ops):
    """stp3_layernorm_fwd / _bwd (stp3/layers/convolutions.py:283-307 LayerNorm over the channels, + the GELU behind it in
    Bottleblock) against float64 torch autograd: float32 and bf16 rows, 32 / 64 / 24-of-a-slice channels, with and without
    the activation; outputs, input gradient, gamma / beta gradients."""
    import torch.nn.functional as F
    from stp3_amd import ops_pred
    out = {}
    g = torch.Generator().manual_seed(21)
    cl = torch.channels_last
    for name, (c, dtype, act, sliced) in {'c64_f32': (64, torch.float32, False, False), 'c32_f32_gelu': (32, torch.float32, True, False),
                                          'c64_bf16': (64, torch.bfloat16, False, False), 'c32_bf16_gelu': (32, torch.bfloat16, True, False),
                                          'c64_bf16_gelu_slice': (64, torch.bfloat16, True, True)}.items():
        n, h, w = 2, 7, 9
        full = (torch.randn(n, c + (16 if sliced else 0), h, w, generator=g) * 1.5 + 0.3).to(dtype).contiguous(memory_format=cl)
        x0 = full[:, 8:8 + c] if sliced else full                       # a channel slice: row stride > C
        w0, b0 = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
        assert ops_pred.layer_norm_supported(x0, c)
        x, wt, b = x0.clone().requires_grad_() if not sliced else x0.detach().requires_grad_(), w0.clone().requires_grad_(), b0.clone().requires_grad_()
        y = ops_pred.layer_norm_channels(x, wt, b, 1e-6, ops_pred.ACT_GELU if act else ops_pred.ACT_NONE)
        gy = torch.randn(y.shape, generator=g).to(dtype)
        y.backward(gy)
        xr, wr, br = x0.detach().double().requires_grad_(), w0.double().requires_grad_(), b0.double().requires_grad_()
        yr = F.layer_norm(xr.permute(0, 2, 3, 1), (c,), wr, br, 1e-6).permute(0, 3, 1, 2)
        if act:
            yr = F.gelu(yr)
        yr.backward(gy.double())
        out[name] = {'y': rel(y.detach().float(), yr.detach()), 'dx': rel(x.grad.float(), xr.grad), 'dw': rel(wt.grad, wr.grad),
                     'db': rel(b.grad, br.grad), 'dtype': str(y.dtype), 'cl': bool(y.is_contiguous(memory_format=cl))}
    return out


def gru_cell(ops):
    """ops_pred.gru_cell (the three convolutions on stp3_conv.hip + stp3_gru_* around them) against float32 torch autograd of
    the reference's cell (stp3/layers/temporal.py:42-56) on the same bf16-representable tensors and weights: new state, input /
    state gradients, every weight and bias gradient; Cx = 32 and 64, with a gate bias offset."""
    import torch.nn as nn
    from stp3_amd import ops_pred
    out = {}
    cl = torch.channels_last
    for name, (cx, c, b0) in {'cx32_c64': (32, 64, 0.0), 'cx64_c64_b0': (64, 64, 0.5)}.items():
        g = torch.Generator().manual_seed(7)
        n, h, w = 2, 6, 9
        convs = [nn.Conv2d(cx + c, c, 3, padding=1) for _ in range(3)]
        with torch.no_grad():
            for m in convs:
                m.weight.copy_((torch.randn(m.weight.shape, generator=g) * 0.05).to(torch.bfloat16).float())
                m.bias.copy_(torch.randn(c, generator=g) * 0.2)
        x0 = torch.randn(n, cx, h, w, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
        s0 = torch.randn(n, c, h, w, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
        x, st = x0.clone().requires_grad_(), s0.clone().requires_grad_()
        assert ops_pred.gru_cell_supported(x, st, *convs)
        y = ops_pred.gru_cell(x, st, *convs, b0)
        gy = torch.randn(y.shape, generator=g).to(torch.bfloat16)
        y.backward(gy)
        got = [p.grad.clone() for m in convs for p in (m.weight, m.bias)]
        for m in convs:
            m.weight.grad = m.bias.grad = None
        xr, sr = x0.float().requires_grad_(), s0.float().requires_grad_()
        xs = torch.cat([xr, sr], 1)
        u = torch.sigmoid(convs[0](xs) + b0)
        r = torch.sigmoid(convs[1](xs) + b0)
        tl = convs[2](torch.cat([xr, (1.0 - r) * sr], 1))
        yr = (1.0 - u) * sr + u * tl
        yr.backward(gy.float())
        want = [p.grad for m in convs for p in (m.weight, m.bias)]
        out[name] = {'y': rel(y.detach().float(), yr.detach()), 'dx': rel(x.grad.float(), xr.grad), 'dstate': rel(st.grad.float(), sr.grad),
                     'dparam': max(rel(a, b) for a, b in zip(got, want)), 'dtype': str(y.dtype)}
    return out


CASES = {f.__name__: f for f in (model_step_f32_full_losses, model_step_two_ranks, lift_full, fuzz, model_step_f32, model_step_f32_bn_eval, model_step_bf16, model_step_bf16_bn_eval, conv_bn, lift_small, lift_c16, lift_c16_rows32, lift_c64_many_runs, lift_c64_rolled, lift_c64_frames, lift_coarse_grid, lift_tall, lift_c64_rows56, voxsum, wprep, optim, se_block, bn_act, bn_act_padded, causal_pair, upsample,
                                 conv, conv_f32, wgrad_defer, assembled_weights, small_linear, fan_out, aspp_join, pointwise_bn, pointwise_stream, decoder_heads, dwconv, layernorm, gru_cell, mbconv_mid, losses, plan, image_prep, labels, bn_group_two_ranks)}

if __name__ == '__main__':
    ops_mod = setup(sys.argv[1])
    t0 = time.time()
    result = CASES[sys.argv[2]](ops_mod)
    result['seconds'] = round(time.time() - t0, 1)
    print('RESULT ' + json.dumps(result))
