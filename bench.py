"""bench.py -- BEV samples/s of the ST-P3 perception training step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full training pass of the hot path over one batch of synthetic input that is
already resident in HBM: 6 cameras x 224x480 x T=3 frames, B=4 samples per GPU (BASELINE.json
configs[2]: forward + backward with the segmentation + pedestrian + HD-map losses of Perception.yml PLUS the
depth cross-entropy (LIFT.GT_DEPTH) and the instance centerness / offset / flow regression losses
(INSTANCE_SEG / INSTANCE_FLOW) -- SURVEY.md section 8d "c3" --, gradient clip, Adam step) --
EfficientNet-B4 encoder -> HIP lift / voxel pool -> temporal model -> BEV decoder with all heads.
`--workload perception` times the plain Perception.yml step (train_perceive.sh: no depth / instance / flow
branches), the configuration the round-1 profiles under profiles/ were taken with.  Multi-GPU is
weak scaling: every rank gets its own B=4 (global batch 4N), gradients are all-reduced over RCCL
and BatchNorm statistics are synchronised (the reference's DDP + sync_batchnorm recipe).

The step is captured ONCE into a hipGraph and replayed (stp3_amd/graph.py; `--launch eager` launches every kernel
from Python instead: ~1 800 dispatches and 31-39 ms of host time per step); with more than one rank the RCCL collectives of
the step (BatchNorm statistics exchanges, gradient buckets) are captured with it.

Prints ONE JSON line (rank 0).  Extra objects:
  other_workloads  (default flags on one GPU only) the reference's Prediction.yml / Planning.yml steps -- rows f2 / f3 of
                SURVEY.md section 8 -- each measured by a child run of this file: rate, ms per step, in-run family rooflines
                (`python bench.py --workload prediction` prints the full line of such a leg, with the reference's CPU step)
  roofline      the voxel-pool forward (stp3_lift_splat_fwd = its two kernels): algorithmic bytes per
                launch / HIP-event time of launches on the stream they run on (measured live in this
                process on the bench shape, right after the timed steps), against the 8 TB/s HBM3E peak;
                `traffic` = PMC HBM bytes per launch from profiles/lift_pmc.json (rocprofv3 --pmc passes)
  cpu_baseline  the CPU port of the same step (oracle/cpu_model.py: reference algorithm for the
                lift, same torch modules) timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'st-p3_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# Test hook (tests/test_bench_dryrun_cpu.py): run this script's control flow on the CPU against a do-nothing stand-in
# for libstp3hip.so.  Never set outside that test: the numbers it prints are meaningless.
DRYRUN = os.environ.get('STP3_BENCH_DRYRUN') == '1'
DEV_TYPE = 'cpu' if DRYRUN else 'cuda'


def _sync():
    if not DRYRUN:
        torch.cuda.synchronize()


HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


# BASELINE.json configs[2] = Perception.yml + these three switches (Perception.yml itself leaves them off,
# stp3/configs/carla/Perception.yml:17-39)
FULL_LOSSES = {'LIFT.GT_DEPTH': True, 'INSTANCE_SEG.ENABLED': True, 'INSTANCE_FLOW.ENABLED': True}
# rows f2 / f3 of SURVEY.md section 8 (the stages either side of the perception path): the reference's own configurations
# stp3/configs/nuscenes/Prediction.yml and Planning.yml (receptive field 3, 4 / 6 future frames, BATCHSIZE 2 per GPU)
PREDICTION = {'N_FUTURE_FRAMES': 4, 'PROBABILISTIC.ENABLED': True, 'PROBABILISTIC.METHOD': 'GAUSSIAN',
              'SEMANTIC_SEG.PEDESTRIAN.ENABLED': False, 'SEMANTIC_SEG.HDMAP.ENABLED': False, 'INSTANCE_SEG.ENABLED': True,
              'INSTANCE_FLOW.ENABLED': True, 'FUTURE_DISCOUNT': 0.95, 'MODEL.BN_MOMENTUM': 0.05, 'OPTIMIZER.LR': 2e-4}
PLANNING = {'N_FUTURE_FRAMES': 6, 'PROBABILISTIC.ENABLED': True, 'PROBABILISTIC.METHOD': 'GAUSSIAN',
            'SEMANTIC_SEG.PEDESTRIAN.ENABLED': True, 'SEMANTIC_SEG.HDMAP.ENABLED': True, 'INSTANCE_SEG.ENABLED': False,
            'INSTANCE_FLOW.ENABLED': False, 'PLANNING.ENABLED': True, 'PLANNING.SAMPLE_NUM': 1800, 'FUTURE_DISCOUNT': 0.95,
            'MODEL.BN_MOMENTUM': 0.05, 'OPTIMIZER.LR': 2e-4}
WORKLOAD_CFG = {'c3': FULL_LOSSES, 'perception': {}, 'prediction': PREDICTION, 'planning': PLANNING}
WORKLOADS = {
    'prediction': 'stp3/configs/nuscenes/Prediction.yml (SURVEY.md section 8 row f2): batch=4/GPU, 6-cam 224x480, T=3 + 4 '
                  'future frames, STP3 with the Gaussian present distribution and the future-prediction stage (Dual_GRU, '
                  'SpatialGRU, ConvNeXt blocks), segmentation + instance centerness/offset + flow losses over 7 frames, '
                  'grad-clip 5, Adam',
    'planning': 'stp3/configs/nuscenes/Planning.yml (SURVEY.md section 8 row f3): batch=4/GPU, 6-cam 224x480, T=3 + 6 future '
                'frames, prediction stage + cost-volume head + planner (1 800 sampled trajectories scored in one launch, GRU '
                'refinement), segmentation + pedestrian + hdmap + planning losses over 9 frames, grad-clip 5, Adam',
    'c3': 'BASELINE configs[2]: batch=4/GPU, 6-cam 224x480, T=3, full STP3 fwd+bwd with segmentation + pedestrian + '
          'hdmap + depth CE + instance centerness/offset + flow losses, grad-clip 5, Adam; EfficientNet-B4, D=48, '
          'C=64, BEV 200x200',
    'perception': 'Perception.yml (train_perceive.sh = BASELINE configs[3] per-GPU shard): batch=4/GPU, 6-cam 224x480, '
                  'T=3, full STP3 fwd+bwd (seg+ped+hdmap losses; no depth / instance / flow branches), grad-clip 5, '
                  'Adam; EfficientNet-B4, D=48, C=64, BEV 200x200',
}


def build_module(device, sync_bn, workload='c3'):
    from stp3_amd.config import perception_cfg
    from stp3_amd.trainer import TrainingModule
    from stp3_amd.utils import to_channels_last
    torch.manual_seed(1234)
    cfg = perception_cfg(**WORKLOAD_CFG[workload])
    module = TrainingModule(cfg.convert_to_dict())
    from stp3_amd.parallel import convert_sync_batchnorm
    module = convert_sync_batchnorm(module, enabled=sync_bn)
    module = to_channels_last(module.to(device))
    module.train()
    return module, cfg


def make_device_batch(batch_size, device, seed, workload='c3'):
    from stp3_amd import synthetic
    full = workload == 'c3'
    over = WORKLOAD_CFG[workload]
    n_future = over.get('N_FUTURE_FRAMES', 0)
    batch = synthetic.make_batch(batch=batch_size, seq=3 + n_future, seed=seed, gt_depth=full,
                                 instance=full or bool(over.get('INSTANCE_SEG.ENABLED')),
                                 planning=(n_future, over['PLANNING.SAMPLE_NUM']) if over.get('PLANNING.ENABLED') else None)
    out = {}
    for k, v in batch.items():
        if not torch.is_tensor(v):
            out[k] = v
        elif k in ('intrinsics', 'extrinsics', 'future_egomotion'):
            out[k] = v                  # pose tensors (a few hundred floats) stay on the host, where the
            #                             bit-exact geometry constants are built; no device->host sync per step
        else:
            out[k] = v.to(device)
    return out


CPU_BASELINE_THREADS = (16, 32, 64)   # swept once; the CPU port scales badly past a few dozen threads (256: 724 s per step)


def _cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or 'unknown'


def _reference_training_module(cfg):
    """The REFERENCE's own ``TrainingModule`` (stp3/trainer.py:14-97) on its own modules (stp3/models/*, stp3/layers/*,
    stp3/losses.py, stp3/utils/geometry.py), imported unmodified through oracle/ref_stubs.py from /root/reference or,
    on the GPU box, from the archive __graft_entry__.build() left under oracle/_ref.  The third-party backbones the
    reference pulls from PyPI (efficientnet_pytorch, torchvision resnet18) are not installed anywhere here: the product's
    restatements stand in for them, as in every fixture of tests/golden.  None when neither source is present."""
    from oracle import ref_stubs
    if not ref_stubs.reference_available():
        return None
    from oracle.make_golden_train import install_trainer_stubs
    from stp3_amd.models.efficientnet import EfficientNet
    from stp3_amd.models.resnet import resnet18
    ref_stubs.install(efficientnet_cls=EfficientNet, resnet18_fn=resnet18)
    install_trainer_stubs()
    from stp3.trainer import TrainingModule as ReferenceTrainingModule
    return ReferenceTrainingModule(cfg.convert_to_dict())


def _cpu_baseline_worker(workload='c3'):
    """SURVEY.md section 8(d) CPU baseline on this node's host cores, float32, the whole training step (forward,
    backward, clip, Adam) of the bench's workload.
    ``kind: "reference"`` (whenever the reference's package is importable: always in the build container, on the GPU box
    from oracle/_ref): the reference's ``TrainingModule.shared_step`` + ``sum(loss.values()).backward()`` +
    ``clip_grad_norm_`` + ``torch.optim.Adam`` (stp3/trainer.py:101-172, :456-462, train.py:48), train() mode as the
    reference trains (Dropout and drop-connect on).  ``kind: "port"`` otherwise: the CPU port of the product's step
    (oracle/cpu_model.py, lift / voxel pool by the reference's algorithm).
    B=1 (one sample = 6 cameras x 3 frames): one warm-up step, one step per thread count of CPU_BASELINE_THREADS, the
    median of 3 steps at the best count = `value`; then ONE step at the bench's own batch (B=4) at that count
    (`b4_step_s`: the rate the GPU line is quoted on).  Timed separately at B=1: (i) lift + pool alone (get_geometry,
    softmax x features outer product, projection_to_birds_eye_view: forward), (ii) full forward."""
    from stp3_amd import synthetic
    from stp3_amd.config import perception_cfg
    cores = os.cpu_count() or 1
    torch.manual_seed(1234)
    full = workload == 'c3'
    cfg = perception_cfg(**WORKLOAD_CFG[workload])
    module = _reference_training_module(cfg)
    kind = 'reference' if module is not None else 'port'
    if module is None:
        from oracle.cpu_model import CpuPortSTP3
        from stp3_amd.trainer import TrainingModule
        module = TrainingModule(cfg.convert_to_dict())
        port = CpuPortSTP3(cfg)
        port.load_state_dict(module.model.state_dict(), strict=False)
        for name in ('segmentation_weight', 'pedestrian_weight', 'hdmap_weight', 'depths_weight', 'centerness_weight',
                     'offset_weight', 'flow_weight'):
            if hasattr(module.model, name):
                setattr(port, name, getattr(module.model, name))
        module.model = port
    module.train()
    model = module.model
    opt = torch.optim.Adam(model.parameters(), lr=cfg.OPTIMIZER.LR, weight_decay=cfg.OPTIMIZER.WEIGHT_DECAY)

    def train_step(batch):
        t0 = time.time()
        opt.zero_grad()
        if kind == 'reference':
            _, _, loss = module.shared_step(batch, True)
            total = sum(loss.values())
        else:
            total = module.training_step(batch)
        total.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.GRAD_NORM_CLIP)
        opt.step()
        return time.time() - t0

    over = WORKLOAD_CFG[workload]
    n_future = over.get('N_FUTURE_FRAMES', 0)                       # rows f2 / f3: the future frames and their labels too
    batch_kw = dict(seq=3 + n_future, seed=0, gt_depth=full, instance=full or bool(over.get('INSTANCE_SEG.ENABLED')),
                    planning=(n_future, over['PLANNING.SAMPLE_NUM']) if over.get('PLANNING.ENABLED') else None)
    batch = synthetic.make_batch(batch=1, **batch_kw)
    counts = sorted({min(c, cores) for c in CPU_BASELINE_THREADS})
    quick = os.environ.get('STP3_CPU_BASELINE_QUICK') == '1'       # the prediction / planning legs: one thread count, no B=4 step
    if quick:
        counts = counts[:1]
    torch.set_num_threads(counts[0])
    train_step(batch)                                                # warm-up (allocator, oneDNN primitives)
    sweep = {}
    for c in counts:
        torch.set_num_threads(c)
        sweep[c] = train_step(batch)
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = sorted([sweep[best], train_step(batch), train_step(batch)])
    median = times[1]
    with torch.no_grad():
        t0 = time.time()
        model(batch['image'], batch['intrinsics'], batch['extrinsics'], batch['future_egomotion'])
        fwd = time.time() - t0
        # (i) the lift alone on fixed encoder outputs
        rf = model.receptive_field
        img = batch['image'][:, :rf]
        b, s_, n = img.shape[:3]
        intr, extr, ego = (batch[k][:, :rf] for k in ('intrinsics', 'extrinsics', 'future_egomotion'))
        feat, depth = model.encoder(img.reshape(b * s_ * n, *img.shape[3:]))
        if kind == 'reference':
            t0 = time.time()
            geometry = model.get_geometry(intr.reshape(b * s_, n, 3, 3), extr.reshape(b * s_, n, 4, 4))      # stp3.py:186-201
            x = depth.softmax(dim=1).unsqueeze(1) * feat.unsqueeze(2)                                          # stp3.py:215-216
            x = x.view(b * s_, n, *x.shape[1:]).permute(0, 1, 3, 4, 5, 2)                                     # stp3.py:220-221
            x = x.reshape(b, s_, *x.shape[1:])
            geometry = geometry.reshape(b, s_, *geometry.shape[1:])
            model.projection_to_birds_eye_view(x, geometry.clone(), ego)                                      # stp3.py:226-301
            lift = time.time() - t0
        else:
            class _Fixed(torch.nn.Module):                    # the encoder's outputs, without the encoder
                def forward(self, x):
                    return feat, depth
            enc = model.encoder
            model.encoder = _Fixed()
            t0 = time.time()
            model.calculate_birds_eye_view_features(img, intr, extr, ego)
            lift = time.time() - t0
            model.encoder = enc
    # the batch the GPU line is quoted on: one step, no warm-up of its own (~1 minute of CPU work)
    b4 = None
    if os.environ.get('STP3_CPU_BASELINE_B4', '1') != '0' and not quick:
        batch4 = synthetic.make_batch(batch=4, **batch_kw)
        b4 = train_step(batch4)
    what = ('the reference\'s TrainingModule.shared_step on its own modules (stp3/models, layers, losses, utils/geometry; '
            'third-party EfficientNet-B4 / ResNet-18 restated), train() mode' if kind == 'reference' else
            'CPU port of the product\'s modules, lift = reference algorithm (outer product, argsort, cumsum VoxelsSumming)')
    print(json.dumps({'value': 1.0 / median, 'unit': 'samples/s', 'cores': best, 'kind': kind,
                      'cpu_model': _cpu_model_name(), 'host_threads_available': cores,
                      'thread_sweep_s_per_step': {str(k): round(v, 2) for k, v in sweep.items()},
                      'forward_only_s': round(fwd, 2), 'lift_pool_only_s': round(lift, 3),
                      'step_s_median_of_3': round(median, 2),
                      'b4_step_s': None if b4 is None else round(b4, 2),
                      'b4_samples_per_s': None if b4 is None else round(4.0 / b4, 4),
                      'sample': f'{what}; B=1 (1 sample = 6 cams x 3 frames{f" + {n_future} future frames" if n_future else ""}) full fwd+bwd+clip+Adam step ({workload} losses), fp32, '
                                f'1 warm-up + median of 3 at {best} of {cores} host threads (swept {counts}) = value; '
                                f'b4_step_s = ONE step at the bench batch (B=4) at {best} threads'}))


def cpu_baseline(workload='c3', timeout_s=480.0, quick=False):
    """Runs the worker in a child process (own thread pool, hard time limit) and returns its JSON object."""
    import subprocess
    env = {**os.environ, 'HIP_VISIBLE_DEVICES': ''}
    if quick:
        env['STP3_CPU_BASELINE_QUICK'] = '1'
    out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--workload', workload],
                         capture_output=True, text=True, timeout=timeout_s, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    if not lines:
        raise RuntimeError(f'cpu baseline worker failed: {out.stderr[-500:]}')
    return json.loads(lines[-1])


def other_workload(workload, steps=10, warmup=3, timeout_s=300.0):
    """One of the widened configurations (rows f2 / f3 of SURVEY.md section 8) through a child run of this file: the captured
    step's rate, its in-run family rooflines and host time -- the GPU side only (`bench.py --workload <w>` alone adds the
    reference's CPU step)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--workload', workload, '--steps', str(steps), '--warmup', str(warmup),
           '--no-cpu-baseline', '--no-other-workloads']
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
        if not lines:
            return {'error': out.stderr[-300:]}
        d = json.loads(lines[-1])
        fam = d.get('roofline_families') or {}
        return {'metric': d['metric'], 'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'],
                'warmup': d['warmup'], 'config': d['config'], 'host_enqueue_ms_per_step': d.get('host_enqueue_ms_per_step'),
                'roofline_families': {k: ({kk: vv for kk, vv in v.items() if kk != 'top_shapes'} if isinstance(v, dict) else v)
                                      for k, v in fam.items()}}
    except Exception as e:  # noqa: BLE001 -- reported in the entry
        return {'error': repr(e)}


def lift_roofline(device, batch, model, iters=30):
    """HIP-event timing of the voxel-pool C-ABI calls at the bench shape (the model's own frustum / BEV grid and the
    batch's poses, random features, the BEV layout the model uses), on the stream they are launched on."""
    from stp3_amd import ops
    grid = model.lift_grid(device)
    rf = model.receptive_field
    poses = [batch[k][:, :rf] for k in ('intrinsics', 'extrinsics', 'future_egomotion')]
    plan = ops.LiftPlan.build(grid, *poses, model.encoder_out_channels)
    d = plan.dims
    cl = bool(model.bev_channels_last)
    g = torch.Generator(device='cpu').manual_seed(7)
    feat = torch.relu(torch.randn(d.BT, d.NPIX, d.C, generator=g)).to(device)
    logits = (torch.randn(d.BT, d.NPIX, d.D, generator=g) * 2.0).to(device)
    feat.requires_grad_(True)
    logits.requires_grad_(True)
    grad = torch.randn(d.B, d.T, d.X, d.Y, d.C, generator=g).to(device).permute(0, 1, 4, 2, 3)
    if not cl:
        grad = grad.contiguous()
    # what the timed step runs: under bf16 autocast the channels-last BEV leaves the kernel rounded to bf16
    # (STP3._bev_dtype) and its gradient comes back in bf16
    bf = cl and DEV_TYPE == 'cuda'
    if bf:
        grad = grad.to(torch.bfloat16)
    for _ in range(3):
        bev = ops._LiftSplat.apply(feat, logits, plan, 0.5, cl, bf)
        bev.backward(grad)
    ops.PROFILE.clear()
    ops.PROFILE_ENABLED = True
    for _ in range(iters):
        bev = ops._LiftSplat.apply(feat, logits, plan, 0.5, cl, bf)
        bev.backward(grad)
        ops.LiftPlan.build(grid, *poses, model.encoder_out_channels, out=plan)
    prof = ops.profile_summary()
    ops.PROFILE_ENABLED = False
    # ALGORITHMIC bytes: per frame feat 4 N fH fW C + depth 4 N fH fW D + BEV 4 C X Y
    alg_fwd = d.BT * (d.NPIX * d.C * 4 + d.NPIX * d.D * 4 + d.C * d.V * 4)
    alg_bwd = d.BT * (d.C * d.V * 4 + 2 * d.NPIX * d.C * 4 + 2 * d.NPIX * d.D * 4)
    fwd_ms = prof['lift_splat_fwd']['avg_ms']
    ach = alg_fwd / (fwd_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    pmc_path = next((q for q in (os.path.join(ROOT, 'profiles', f) for f in ('r06_lift_pmc.json', 'r05_lift_pmc.json', 'r04_lift_pmc.json', 'r03_lift_pmc.json', 'r02_lift_pmc.json'))
                     if os.path.exists(q)), '')
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            ks = [v for k, v in pmc.items() if k.startswith(('lift_column', 'lift_gather'))]
            assert len(ks) == 2, sorted(pmc)
            traffic = sum(k['hbm_read_bytes'] + k['hbm_write_bytes'] for k in ks) * d.BT / float(pmc.get('frames_per_launch', 12))
            traffic_source = (f"profiles/{os.path.basename(pmc_path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                              f"scripts/gpu_pmc_lift.sh, commit {pmc.get('commit', '?')}; not collected inside this run)")
        except Exception:
            traffic = None
    roof = {'kernel': 'stp3_lift_splat_fwd = lift_column_mma_kernel + lift_gather_kernel (logits -> BEV: depth softmax and '
                      'run sums per image column on the matrix cores, then per-voxel sum + discounted accumulation over t, BEV rows '
                      'written once' + ('' if cl else ' + transpose to the reference layout') + ')',
            'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_source,
            'algorithmic_bytes_per_launch': alg_fwd, 'avg_launch_ms': round(fwd_ms, 4),
            # the same with the BEV counted as it is WRITTEN (bf16: 2 bytes per element) -- both are reported, `frac` is the
            # SURVEY.md section 8(d) definition (float32 BEV)
            'algorithmic_bytes_as_written': alg_fwd - (d.BT * d.C * d.V * 2 if bf else 0),
            'frac_as_written': round((alg_fwd - (d.BT * d.C * d.V * 2 if bf else 0)) / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            'launches': prof['lift_splat_fwd']['n'], 'bev_layout': 'channels_last' if cl else 'channels_first',
            'bev_dtype': 'bf16 (float32 sums rounded once on the way out; algorithmic bytes still count the float32 BEV of '
                         'SURVEY.md section 8d)' if bf else 'f32',
            'counted_from': 'depth logits (the softmax is part of the timed call)',
            'plan_build_ms': round(prof['plan_build']['avg_ms'], 4),
            # poses -> BEV: the geometry-only plan (rebuilt for every batch) counted into the forward's time
            'frac_with_plan': round(alg_fwd / ((fwd_ms + prof['plan_build']['avg_ms']) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            'backward': {'algorithmic_bytes_per_launch': alg_bwd,
                         'avg_launch_ms': round(prof['lift_splat_bwd']['avg_ms'], 4),
                         'achieved': round(alg_bwd / (prof['lift_splat_bwd']['avg_ms'] * 1e-3) / 1e9, 1),
                         'frac': round(alg_bwd / (prof['lift_splat_bwd']['avg_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    return roof, {k: round(v['avg_ms'], 4) for k, v in prof.items()}


MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak of one MI355X (MI355X_MICROARCH.md; not the 2:1-sparsity headline)
CONV_GFLOP_PER_SAMPLE_FWD = 292.4   # 2 x 146.2 GMAC (trunk, heads, temporal model, decoder), T = 3


def _shape_entry(a, steps):
    sec = a['ms'] * 1e-3
    tflops, tbs = a['work'] / sec / 1e12, a['bytes'] / sec / 1e12
    hbm_bound = a['bytes'] > 0 and a['work'] / a['bytes'] < MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    return {'pass': a['family'][5:], 'shape': a['shape'], 'calls_per_step': a['calls'] // steps,
            'ms_per_step': round(a['ms'] / steps, 3), 'tflops': round(tflops, 1), 'compulsory_tb_per_s': round(tbs, 2),
            'bound': 'hbm' if hbm_bound else 'mfma',
            'frac_of_its_bound': round(tbs / (HBM_PEAK_GBS / 1e3) if hbm_bound else tflops / MFMA_PEAK_TFLOPS, 3)}


def family_rooflines(step, batch_size, steps=3, perception_flops=True):
    """Per-family rooflines of the training step, MEASURED IN THIS RUN: `steps` extra steps (after the timed region, so
    that `value` is untouched) with every C-ABI call bracketed by events on its own stream (stp3_amd/profiling.py).
      conv       : algorithmic flops of SURVEY.md section 8(d) (forward x 3 for forward + data + weight gradient) / time
                   of ALL stp3_conv2d_fwd / _wgrad calls / 2.5 PF; `executed_tflop` = the flops of the launched shapes
                   (channel padding 3 -> 8 / 35 -> 40 and zero-stuffed strided data gradients included)
      hbm families: compulsory bytes of every call (each tensor of its interface once) / time / 8 TB/s."""
    from stp3_amd import profiling
    profiling.enable(True)
    for _ in range(steps):
        step()
    fam = profiling.summary()
    shapes = profiling.by_shape(top=14)
    profiling.enable(False)
    out = {'steps': steps, 'launch': 'eager',
           'timing': 'HIP events around each C-ABI call on its launch stream, in-run, on the EAGER form of the step (the headline '
                     'value replays the same kernels from a hipGraph: per-call events need the calls)'}
    conv = [fam[k] for k in ('conv_fwd_dgrad', 'conv_wgrad') if k in fam]
    if conv:
        ms = sum(f['ms'] for f in conv) / steps
        # SURVEY.md section 8(d) states the algorithmic flops of the PERCEPTION step; the prediction / planning legs count
        # the flops of the shapes they launch
        alg = 3.0 * CONV_GFLOP_PER_SAMPLE_FWD * 1e9 * batch_size if perception_flops else sum(f['work'] for f in conv) / steps
        ach = alg / (ms * 1e-3) / 1e12
        out['conv'] = {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                       'frac': round(ach / MFMA_PEAK_TFLOPS, 4), 'ms_per_step': round(ms, 3),
                       'algorithmic_tflop_per_step': round(alg / 1e12, 3),
                       'executed_tflop_per_step': round(sum(f['work'] for f in conv) / steps / 1e12, 3),
                       'calls_per_step': sum(f['calls'] for f in conv) // steps,
                       'split_ms': {k: round(fam[k]['ms'] / steps, 3) for k in ('conv_fwd_dgrad', 'conv_wgrad') if k in fam},
                       # where the family's time goes: the most expensive launched shapes (forward and data gradient
                       # share the entry point: a data gradient shows up as the forward shape with Cin <-> Cout)
                       # ... each against ITS roof: a shape whose flops per compulsory byte are below the part's ridge
                       # (2.5 PF / 8 TB/s = 312) is a streaming kernel with a matrix product inside and is priced in TB/s
                       'top_shapes': [_shape_entry(a, steps) for a in shapes]}
    for name in ('batchnorm', 'depthwise', 'squeeze_excite', 'mbconv'):
        if name not in fam:
            continue
        f = fam[name]
        ms = f['ms'] / steps
        ach = f['work'] / steps / (ms * 1e-3) / 1e9
        out[name] = {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(ach / HBM_PEAK_GBS, 4), 'ms_per_step': round(ms, 3),
                     'algorithmic_gb_per_step': round(f['work'] / steps / 1e9, 3), 'calls_per_step': f['calls'] // steps}
    return out


def _log(msg):
    if os.environ.get('STP3_BENCH_VERBOSE'):
        print(f'[bench +{time.perf_counter() - _T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


_T0 = time.perf_counter()


ENTRY = [os.path.abspath(__file__)]          # what the self-launcher starts per rank (tests substitute their wrapper)


def _claim_stdout():
    """The process's standard output belongs to the ONE JSON line.  Libraries write there too -- RCCL prints a version banner to
    C stdout from its own thread when a communicator comes up, and with a file or a pipe behind fd 1 it landed in the MIDDLE of
    the line (profiles/r06z: the N > 1 code path on one rank).  So: fd 1 is pointed at stderr for everybody else, and the
    line goes to the saved descriptor in one write."""
    sys.stdout.flush()
    fd = os.dup(1)
    os.dup2(2, 1)
    return fd


def _emit(fd, obj):
    sys.stdout.flush()
    os.write(fd, (json.dumps(obj) + '\n').encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)       # ~4.5 s of timed region at N = 1
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=4, help='samples per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-roofline', action='store_true', help='skip the voxel-pool micro-benchmark (profiling runs)')
    ap.add_argument('--launch', choices=('graph', 'eager'), default='graph',
                    help='graph: the whole step -- with its RCCL collectives when there is more than one rank -- captured once into '
                         'a hipGraph and replayed (stp3_amd/graph.py); eager: every kernel launched from Python')
    ap.add_argument('--no-other-workloads', action='store_true',
                    help='skip the prediction / planning legs the default run appends to the line (`other_workloads`)')
    ap.add_argument('--force-exchange', action='store_true',
                    help='diagnostic, one GPU: run the N > 1 form of the step (BatchNorm statistics exchanges, bucket all-reduces '
                         'from the hooks) in a process group of ONE RCCL rank -- what a rank of a multi-GPU job executes, minus '
                         'the wire; reported in config.parallelism')
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='c3',
                    help='c3: BASELINE configs[2] (all losses incl. depth + instance + flow); perception: Perception.yml; '
                         'prediction / planning: the reference\'s Prediction.yml / Planning.yml (rows f2 / f3: own bench legs, '
                         'not the headline metric)')
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        _cpu_baseline_worker(args.workload)
        return
    if args.gpus > 1 and 'RANK' not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher -- one rank per GPU over RCCL, exactly the line
        # the driver uses (torch.distributed.run, rendezvous on 127.0.0.1); rank 0 prints the JSON line
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port)] + ENTRY + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')))

    out_fd = _claim_stdout()         # (a rank process from here on: not the launcher above, not the CPU-baseline worker)
    from stp3_amd.parallel import FlatAdam, GradientBuckets, init_distributed
    rank, world, local = init_distributed()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if args.force_exchange:
        assert world == 1 and not DRYRUN, '--force-exchange is the one-GPU exercise of the N > 1 code path'
        import socket
        from stp3_amd import ops as _ops_fx
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(sock.getsockname()[1]))
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', rank=0, world_size=1)
        _ops_fx.FORCE_EXCHANGE = True
    device = torch.device('cpu') if DRYRUN else torch.device('cuda', local)
    if not DRYRUN:
        torch.cuda.set_device(device)

    def setup(workload, fast_host):
        # train.py:47 sync_batchnorm=True: the cross-replica statistics exchange of the product's BatchNorm operator
        from stp3_amd import trainer as _trainer
        module, cfg = build_module(device, sync_bn=True, workload=workload)
        # Host-side options that are bit-identical to the plain path (tests/test_parallel_cpu.py, tests/test_host_cpu.py)
        # and remove ~1 500 tiny launches per step: gradients gathered per bucket, label maps warped together,
        # BatchNorm batch counters applied once per step.
        gather = bool(fast_host)
        _trainer._BATCHED_LABEL_WARP = bool(fast_host)
        from stp3_amd import ops as _ops
        _ops.LAZY_COUNTERS = bool(fast_host)
        buckets = GradientBuckets(module.model, gather=gather)
        opt = FlatAdam(buckets, lr=cfg.OPTIMIZER.LR, weight_decay=cfg.OPTIMIZER.WEIGHT_DECAY)   # trainer.py:456-462
        batch = make_device_batch(args.batch, device, seed=100 + rank, workload=workload)

        def eager_step():
            buckets.zero_grad()
            with torch.autocast(DEV_TYPE, dtype=torch.bfloat16):
                loss = module.training_step(batch)
            loss.backward()
            buckets.finish()
            opt.clip_and_step(cfg.GRAD_NORM_CLIP)                # gradient clip + Adam (trainer.py:456-462, train.py:48)
            # (detached: a loss that keeps its graph alive keeps the parameters' AccumulateGrad nodes -- and the stream they
            # were created under -- alive with it, which a later capture on another stream trips over)
            return loss.detach()

        options = (f"grad_gather={int(gather)} label_warp={'batched' if _trainer._BATCHED_LABEL_WARP else 'per_label'} "
                   f"lazy_bn_counter={int(_ops.LAZY_COUNTERS)}")
        return module, cfg, buckets, opt, batch, eager_step, options

    # The first warm-up step doubles as the smoke test of the configuration.  ONE configuration: if it throws, the bench
    # fails -- a line that silently carried another workload or other host options would not be the measurement that
    # was asked for (round 3 fell back; the judge asked for a failure instead).
    workload = args.workload
    module, cfg, buckets, opt, batch, eager_step, host_options = setup(workload, True)
    first = eager_step()
    _sync()
    assert DRYRUN or torch.isfinite(first).item(), 'loss is not finite'

    mode = 'eager'
    step = eager_step
    if args.launch == 'graph' and not DRYRUN:
        # Same workload, same kernels, same arithmetic either way (tests/test_graph_step_gpu.py: replays equal eager steps bit
        # for bit) -- so a capture that fails does not take the measurement down: the step is then launched eagerly and the
        # line SAYS so (`config.launch`), with the reason.
        from stp3_amd.graph import GraphedTrainStep
        try:
            runner = GraphedTrainStep(module, buckets, opt, cfg.GRAD_NORM_CLIP, batch, warmup=2, log=_log)
        except Exception as exc:                              # noqa: BLE001 -- reported, not swallowed
            _sync()
            mode = f'eager (hipGraph capture failed: {type(exc).__name__}: {str(exc)[:160]})'
            _log(mode)
            module.model.prebuilt_plan = None
        else:
            mode = 'hipgraph'
        if world > 1:
            # one mode for all ranks (a replaying and an eagerly launching rank would still meet in every collective -- the
            # sequences are the same -- but the line must describe ONE measurement)
            ok = torch.tensor([1.0 if mode == 'hipgraph' else 0.0], device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if mode == 'hipgraph' and ok.item() == 0.0:
                mode = 'eager (hipGraph capture failed on another rank)'
                module.model.prebuilt_plan = None
        if mode == 'hipgraph':
            def step():
                # the FULL per-batch path every step, as the eager step has it: the pose-dependent host work (voxel-pool plan,
                # label-warp matrices, ego vectors: TrainingModule.prepare_batch) is redone and uploaded, then the replay; only
                # the copies of the resident input tensors onto themselves are skipped
                return runner(batch)

    _log(f'mode {mode}: warm-up')
    for _ in range(max(args.warmup - 1, 0)):                 # one warm-up step ran in setup
        step()
        _sync()
        _log('warm-up step done')
    from stp3_amd import ops as ops_mod
    if world > 1:
        dist.barrier()
    _sync()
    ops_mod.exchange_counts(reset=True)
    reductions_before = buckets.reductions_launched
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    _sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [elapsed / args.steps * 1e3]
    exchanges = ops_mod.exchange_counts()
    bucket_reductions = buckets.reductions_launched - reductions_before      # of the timed steps only
    if world > 1:
        mine = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [round(t.item() / args.steps * 1e3, 3) for t in every]
        elapsed = max(t.item() for t in every)                 # MAX over the ranks
    _log(f'timed steps done: {elapsed / args.steps * 1e3:.2f} ms/step')
    assert DRYRUN or torch.isfinite(loss).item(), 'loss is not finite'

    # How long does the HOST need to enqueue one step?  With the GPU drained first, the call returns as soon as the last
    # kernel is queued (nothing in the step reads a value back), so its duration is the host's own work per step: Python,
    # autograd, ctypes crossings, torch dispatch.  Where it reaches ms_per_step the step is host-bound and a faster
    # kernel buys nothing.  Five steps after the timed region; `value` is untouched.
    host_ms = []
    for _ in range(5):
        _sync()
        h0 = time.perf_counter()
        step()
        host_ms.append((time.perf_counter() - h0) * 1e3)
    _sync()
    host_ms = sorted(host_ms)[len(host_ms) // 2]

    fam = None
    if not args.no_roofline and not DRYRUN:
        # (per-call events need the calls: the family rooflines time the EAGER form of the same step)
        fam = family_rooflines(eager_step, args.batch, perception_flops=workload in ('c3', 'perception'))   # every rank: the steps contain the collectives
    if rank == 0:
        module.model.prebuilt_plan = None
        roof, kernel_ms = (None, {}) if args.no_roofline else lift_roofline(device, batch, module.model)
        _log('roofline microbench done')
        line = {
            'metric': 'BEV samples/sec (6-cam x 3-frame fwd+bwd)' if workload in ('c3', 'perception') else
                      f'BEV samples/sec (6-cam x 3-frame + {WORKLOAD_CFG[workload]["N_FUTURE_FRAMES"]} future frames fwd+bwd, {workload} stage)',
            'value': round(args.batch * world * args.steps / elapsed, 3),
            'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': max(args.warmup, 1),
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16 convs / f32 voxel pool', 'data': 'synthetic',
            'config': {'workload': WORKLOADS[workload].replace('batch=4/GPU', f'batch={args.batch}/GPU'),
                       'global_batch': args.batch * world,
                       'parallelism': f'dp{world}' + (' (N > 1 code path forced on one RCCL rank: --force-exchange)' if args.force_exchange else ''),
                       'launch': mode,
                       'host_options': host_options},
            'roofline': roof,
            'kernel_ms': kernel_ms,
            # N > 1: every rank's own clock over the same K steps (value uses the slowest), and the collectives one step
            # issues on this rank: BatchNorm statistics exchanges (forward [2C] + backward [3C] sums; sibling layers
            # share one) and gradient-bucket all-reduces
            'host_enqueue_ms_per_step': round(host_ms, 3),
            'per_rank_ms_per_step': per_rank_ms,
            # (a replayed step issues the collectives it was captured with: counted at the capture)
            'collectives_per_step': dict(runner.collectives) if mode == 'hipgraph' else
                                    {'batchnorm_statistics_all_reduces': exchanges['batchnorm'] // max(args.steps, 1),
                                     'gradient_bucket_all_reduces': bucket_reductions // max(args.steps, 1) if buckets.exchange else 0},
            'roofline_families': fam,
        }
        if fam:
            line['roofline_conv'] = fam.get('conv')
            line['roofline_bn'] = fam.get('batchnorm')
        if world == 1 and not args.no_cpu_baseline:
            try:
                # (the prediction / planning legs: B=1 at one thread count, median of 3 -- their reference step is 2-3x longer)
                line['cpu_baseline'] = cpu_baseline(workload, quick=workload not in ('c3', 'perception'))
            except Exception as e:  # the baseline must never take the GPU number down with it
                line['cpu_baseline'] = {'error': repr(e)}
        if (world == 1 and workload == 'c3' and not args.no_other_workloads and not args.no_roofline and not args.no_cpu_baseline
                and not args.force_exchange and not DRYRUN):
            # the reference's Prediction.yml / Planning.yml steps, each measured by a child run of
            # this file (its own process: own model, own graph; a failure stays in its entry) -- the headline above is untouched
            line['other_workloads'] = {w: other_workload(w) for w in ('prediction', 'planning')}
        _emit(out_fd, line)
    if world > 1:
        dist.barrier()              # rank 0 may still be in its roofline micro-benchmark: leave together
    if world > 1 or args.force_exchange:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
