"""Bisect hipGraph capture of the training step: python scripts/graph_probe.py <stage>
stages: encoder | lift | temporal | decoder | loss | optim | full"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import torch
from stp3_amd import synthetic
from stp3_amd.config import perception_cfg
from stp3_amd.trainer import TrainingModule
from stp3_amd.utils import to_channels_last
from stp3_amd.parallel import FlatAdam, GradientBuckets

stage = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda', 0)
torch.manual_seed(1234)
cfg = perception_cfg()
module = to_channels_last(TrainingModule(cfg.convert_to_dict()).to(dev)).train()
model = module.model
batch = synthetic.make_batch(batch=B, seq=3, seed=1)
dbatch = {k: (v.to(dev) if torch.is_tensor(v) and k not in ('intrinsics', 'extrinsics') else v) for k, v in batch.items()}


def log(m):
    print(f'[{stage}] {m}', flush=True)


def snapshot():
    snap = {'param:' + k: v.detach().clone() for k, v in module.named_parameters()}
    snap.update({'buf:' + k: v.detach().clone() for k, v in module.named_buffers()})
    return snap


def diff(snap, tag):
    cur = {'param:' + k: v for k, v in module.named_parameters()}
    cur.update({'buf:' + k: v for k, v in module.named_buffers()})
    for k, v in cur.items():
        if 'running_' in k or 'num_batches' in k:
            continue
        if not torch.equal(v, snap[k]):
            d = (v.float() - snap[k].float()).abs()
            log(f'{tag}: CHANGED {k} shape={tuple(v.shape)} ptr={v.data_ptr():#x} n_changed={int((d > 0).sum())} max={float(d.max()):.4g} '
                f'first_idx={int((d.reshape(-1) > 0).nonzero()[0])}')


def run(body, n_replay=3):
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    cur.wait_stream(side)
    torch.cuda.synchronize()
    log('eager warm-up ok')
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body()
    torch.cuda.synchronize()
    log('captured')
    snap = snapshot()
    for i in range(n_replay):
        g.replay()
        torch.cuda.synchronize()
        log(f'replay {i} ok, out={float(out):.5f}')
        diff(snap, f'after replay {i}')


img = dbatch['image']
b, s, n, c, h, w = img.shape
params = [p for p in model.parameters()]
if stage == 'encoder':
    x = img.reshape(b * s * n, c, h, w)
    def body():
        for p in params: p.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            f, d = model.encoder(x)
            loss = f.float().mean() + d.float().mean()
        loss.backward()
        return loss.detach()
    run(body)
elif stage == 'lift':
    from stp3_amd import ops
    plan = model.prepare_plan(batch['intrinsics'], batch['extrinsics'], batch['future_egomotion'], dev)
    d = plan.dims
    feat = torch.rand(b, s, n, 64, 28, 60, device=dev, requires_grad=True)
    logit = torch.randn(b, s, n, 48, 28, 60, device=dev, requires_grad=True)
    def body():
        feat.grad = None; logit.grad = None
        bev = ops.lift_splat(feat, logit, plan, 0.5)
        loss = bev.square().mean()
        loss.backward()
        return loss.detach()
    run(body)
elif stage in ('temporal', 'decoder'):
    x = torch.randn(b, 3, 70 if stage == 'temporal' else 64, 200, 200, device=dev)
    mod = model.temporal_model if stage == 'temporal' else model.decoder
    def body():
        for p in params: p.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            o = mod(x)
            if isinstance(o, dict):
                loss = sum(v.float().mean() for v in o.values() if v is not None)
            else:
                loss = o.float().mean()
        loss.backward()
        return loss.detach()
    run(body)
elif stage in ('temporal_copy', 'temporal_fwd', 'temporal_nodrop', 'temporal_nopyr'):
    tm = model.temporal_model
    xs = torch.randn(b, 3, 70, 200, 200, device=dev)
    if stage == 'temporal_nodrop':
        tm.final_conv[0].project[3].p = 0.0
    if stage == 'temporal_nopyr':
        for blk in tm.model: blk.use_pyramid_pooling = False
    def body():
        for p in params: p.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            xin = xs.clone() if stage == 'temporal_copy' else xs
            if stage == 'temporal_fwd':
                with torch.no_grad():
                    return tm(xin).float().mean()
            o = tm(xin)
            loss = o.float().mean()
        loss.backward()
        return loss.detach()
    run(body)
elif stage == 'temporal_trace':
    from stp3_amd.layers.fused import bn_act, conv_module, run_fused, ACT_RELU
    tm = model.temporal_model
    xs = torch.randn(b, 3, 70, 200, 200, device=dev)
    stats = torch.zeros(12, device=dev)
    def body():
        with torch.autocast('cuda', dtype=torch.bfloat16), torch.no_grad():
            x = tm.model(xs.permute(0, 2, 1, 3, 4)).permute(0, 2, 1, 3, 4)
            bb, ss, cc, hh, ww = x.shape
            x = x.reshape(bb * ss, cc, hh, ww)
            stats[0] = x.float().abs().mean()
            head = tm.final_conv
            aspp = head[0]
            branches = [run_fused(aspp.convs[0], x)] + [conv(x) for conv in aspp.convs[1:-1]]
            spatial = torch.cat(branches, dim=1)
            stats[1] = spatial.float().abs().mean()
            pooled = aspp.convs[-1](x)
            stats[2] = pooled.float().abs().mean()
            proj, bn, act, drop = aspp.project
            n_sp = spatial.shape[1]
            from stp3_amd.layers.fused import conv2d
            y = conv2d(spatial, proj.weight[:, :n_sp])
            stats[3] = y.float().abs().mean()
            sbias = torch.nn.functional.conv2d(pooled.to(y.dtype), proj.weight[:, n_sp:]).flatten(1).float()
            stats[4] = sbias.abs().mean()
            y = bn_act(bn, y, ACT_RELU, sbias=sbias)
            stats[5] = y.float().abs().mean()
            y = drop(y)
            stats[6] = y.float().abs().mean()
            y = conv_module(head[1], y)
            stats[7] = y.float().abs().mean()
            y = bn_act(head[2], y, ACT_RELU)
            stats[8] = y.float().abs().mean()
            y = conv_module(head[4], y)
            stats[9] = y.float().abs().mean()
            stats[10] = head[4].weight.float().abs().mean()
            stats[11] = head[2].weight.float().abs().mean()
            return y.float().mean()
    cur = torch.cuda.current_stream(); side = torch.cuda.Stream(); side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(3): o = body()
    cur.wait_stream(side); torch.cuda.synchronize()
    log(f'eager out={float(o):.5f} stats={[round(float(v), 4) for v in stats]}')
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o = body()
    for i in range(3):
        g.replay(); torch.cuda.synchronize()
        log(f'replay {i} out={float(o):.5f} stats={[round(float(v), 4) for v in stats]}')
elif stage in ('tblock', 'tfinal', 'pyramid', 'padconv', 'dilated'):
    from stp3_amd.layers import temporal as T
    from stp3_amd.layers.fused import bn_act, ACT_RELU
    tm = model.temporal_model
    if stage == 'tblock':
        x = torch.randn(b, 70, 3, 200, 200, device=dev)
        fn = lambda: tm.model(x)
    elif stage == 'tfinal':
        x = torch.randn(b * 3, 64, 200, 200, device=dev)
        fn = lambda: tm.final_conv(x)
    elif stage == 'pyramid':
        x = torch.randn(b, 70, 3, 200, 200, device=dev)
        fn = lambda: sum(o.float().sum() for o in tm.model[0].pyramid_pooling(x))
    elif stage == 'padconv':
        x = torch.randn(b * 3, 70, 200, 200, device=dev).contiguous(memory_format=torch.channels_last)
        blk = tm.model[0]
        fn = lambda: blk._pointwise(blk.convolution_paths[0][0], x)
    else:
        x = torch.randn(b * 3, 64, 200, 200, device=dev)
        fn = lambda: tm.final_conv[0].convs[1](x)
    def body():
        for p in params: p.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            o = fn()
            loss = o.float().mean()
        loss.backward()
        return loss.detach()
    run(body)
elif stage == 'loss':
    out = {'segmentation': torch.randn(b, 3, 2, 200, 200, device=dev, requires_grad=True),
           'pedestrian': torch.randn(b, 3, 2, 200, 200, device=dev, requires_grad=True),
           'hdmap': torch.randn(b, 4, 200, 200, device=dev, requires_grad=True)}
    def body():
        labels = module.prepare_future_labels(dbatch)
        l = module.losses_fn['segmentation'](out['segmentation'], labels['segmentation'], 3)
        l = l + module.losses_fn['pedestrian'](out['pedestrian'], labels['pedestrian'], 3)
        l = l + module.losses_fn['hdmap'](out['hdmap'], labels['hdmap'])
        l.backward()
        return l.detach()
    run(body)
elif stage == 'optim':
    buckets = GradientBuckets(model)
    opt = FlatAdam(buckets, lr=1e-3, weight_decay=1e-7)
    def body():
        buckets.zero_grad()
        for f, _ in buckets.buckets: f.add_(0.01)
        buckets.finish()
        n = buckets.clip_grad_norm_(5.0)
        opt.step()
        return n.detach()
    run(body)
elif stage == 'full':
    from stp3_amd.graph import GraphedTrainStep
    buckets = GradientBuckets(model)
    opt = FlatAdam(buckets, lr=1e-3, weight_decay=1e-7)
    runner = GraphedTrainStep(module, buckets, opt, 5.0, dbatch, log=log)
    for i in range(3):
        loss = runner(dbatch)
        torch.cuda.synchronize()
        log(f'replay {i} ok loss={float(loss):.5f}')
print(f'[{stage}] PASS', flush=True)
