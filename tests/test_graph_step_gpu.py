"""GPU: the training step as a hipGraph (stp3_amd/graph.py) against the eager step it was captured from.

Two identical setups of bench.py's configs[2] step (same weights, same batch, Dropout p = 0 and drop-connect 0 so that no
random stream is involved): one runs W + N eager steps, the other W eager warm-up steps inside ``GraphedTrainStep`` and
then N replays.  The loss of every replayed step, every parameter, every BatchNorm running statistic and batch counter and
the optimizer's step count must equal the eager run's BIT FOR BIT: same kernels, same order, same summation orders -- the
capture changes who issues the launches and nothing else (reference semantics: stp3/trainer.py:101-172, :456-462,
train.py:48)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
C3 = {'LIFT.GT_DEPTH': True, 'INSTANCE_SEG.ENABLED': True, 'INSTANCE_FLOW.ENABLED': True}
WARMUP, STEPS = 2, 4


def _setup(batch_size=2):
    from stp3_amd import synthetic
    from stp3_amd.config import perception_cfg
    from stp3_amd.parallel import FlatAdam, GradientBuckets
    from stp3_amd.trainer import TrainingModule
    from stp3_amd.utils import to_channels_last
    torch.manual_seed(1234)
    cfg = perception_cfg(**C3)
    module = to_channels_last(TrainingModule(cfg.convert_to_dict()).cuda())
    module.train()
    for m in module.modules():                                   # no random streams: the comparison is bitwise
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        gp = getattr(m, '_global_params', None)
        if gp is not None and hasattr(gp, 'drop_connect_rate'):
            gp.drop_connect_rate = 0.0
    buckets = GradientBuckets(module.model, gather=True)
    opt = FlatAdam(buckets, lr=cfg.OPTIMIZER.LR, weight_decay=cfg.OPTIMIZER.WEIGHT_DECAY)
    raw = synthetic.make_batch(batch=batch_size, seq=3, seed=7, gt_depth=True, instance=True)
    batch = {k: (v.cuda() if torch.is_tensor(v) and k not in ('intrinsics', 'extrinsics', 'future_egomotion') else v)
             for k, v in raw.items()}
    return module, cfg, buckets, opt, batch


def _eager_step(module, cfg, buckets, opt, batch):
    buckets.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = module.training_step(batch)
    loss.backward()
    buckets.finish()
    opt.clip_and_step(cfg.GRAD_NORM_CLIP)
    return loss.detach().clone()


def test_replays_equal_the_eager_steps_bit_for_bit():
    from stp3_amd import ops
    from stp3_amd.graph import GraphedTrainStep
    module, cfg, buckets, opt, batch = _setup()
    eager = [_eager_step(module, cfg, buckets, opt, batch) for _ in range(WARMUP + STEPS)]
    ops.flush_batch_counters()
    want_state = {k: v.detach().clone() for k, v in module.state_dict().items()}
    want_steps = opt.step_count
    del module, buckets, opt
    torch.cuda.synchronize()

    module, cfg, buckets, opt, batch = _setup()
    runner = GraphedTrainStep(module, buckets, opt, cfg.GRAD_NORM_CLIP, batch, warmup=WARMUP)
    # (the full per-batch path, as bench.py runs it: plan and warp matrices rebuilt and uploaded before every replay)
    got = [runner(batch).clone() for _ in range(STEPS)]
    torch.cuda.synchronize()
    for i, (g, e) in enumerate(zip(got, eager[WARMUP:])):
        assert torch.isfinite(g).item() and torch.equal(g, e), (i, g.item(), e.item())
    assert opt.step_count == want_steps == WARMUP + STEPS
    assert module.training_step_count == WARMUP + STEPS
    ops.flush_batch_counters()
    state = module.state_dict()
    diff = [k for k, v in want_state.items() if not torch.equal(state[k], v)]
    assert not diff, diff[:8]
    # a NEW batch through the same graph: inputs copied into the static buffers, plan and warp matrices rebuilt in place
    from stp3_amd import synthetic
    raw = synthetic.make_batch(batch=2, seq=3, seed=8, gt_depth=True, instance=True)
    other = {k: (v.cuda() if torch.is_tensor(v) and k not in ('intrinsics', 'extrinsics', 'future_egomotion') else v)
             for k, v in raw.items()}
    on_graph = runner(other).clone()
    module2, cfg2, buckets2, opt2, _ = _setup()
    module2.load_state_dict(want_state)
    # (the eager twin continues from the same parameters but with fresh Adam moments: compare the LOSS of the step only,
    # which depends on parameters and batch alone)
    buckets2.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        want = module2.training_step(other).detach()
    assert torch.equal(on_graph, want), (on_graph.item(), want.item())
