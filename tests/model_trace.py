"""CPU test infrastructure: dry run of ONE TRAINING STEP through the GPU code path, without a GPU.

Driver for tests/test_model_dryrun_cpu.py.  It loads the recording stand-in for libstp3hip.so (tests/host_trace.py),
makes every tensor claim ``is_cuda`` and autocast claim to be on (the real autocast runs for the CPU type, so the
torch operators between the custom ones see the dtypes they would see on the GPU), and runs
``TrainingModule.training_step`` + backward of a small configuration.  The custom kernels do nothing, so values
are meaningless -- what the run establishes is that the host code of the selected path (default or the
experimental switches given in the environment) executes end to end, and which C-ABI calls it makes.

    STP3_TRACE_LOG=... STP3_REAL_LIB=.../libstp3hip.so [STP3_CONV_V2=1 ...] python tests/model_trace.py recorder.so
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def patch_process(recorder, deterministic_fill=True):
    """Make this process run the GPU code path on CPU tensors against the recording library."""
    import contextlib
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
    from stp3_amd import _lib
    _lib.LIB_PATH = recorder
    from stp3_amd import ops
    ops._need_gpu = lambda *a: None
    ops._stream = lambda: None
    ops._stream_handle = lambda: 0
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.is_autocast_enabled = lambda *a: True
    torch.get_autocast_gpu_dtype = lambda: torch.bfloat16
    torch.get_autocast_dtype = lambda *a: torch.bfloat16

    class _Stream:                               # the model overlaps the plan build on a side stream
        cuda_stream = 0

        def wait_stream(self, other):
            pass

    class _Event:                                # bench.py times the voxel-pool launches with events
        def __init__(self, *a, **k):
            pass

        def record(self, *a):
            pass

        def synchronize(self):
            pass

        def elapsed_time(self, other):
            return 0.125

    torch.cuda.Event = _Event
    torch.Tensor.record_stream = lambda self, stream: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.Stream = lambda *a, **k: _Stream()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.set_num_threads(4)
    if deterministic_fill:
        # uninitialised memory (what the do-nothing kernels leave in their outputs) becomes a fixed NaN / max-int
        # pattern, so the buffer checksums in the trace are reproducible from run to run and from path to path
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True


def dry_setup(recorder, final_dim=(64, 96), batch_size=2, bev_cells=64, deterministic_fill=True, full_losses=None):
    """Patch the process for a dry run and build (module, batch, cfg).  Also used by scripts/host_step_profile.py."""
    import torch
    patch_process(recorder, deterministic_fill)

    from stp3_amd import synthetic
    from stp3_amd.config import perception_cfg
    from stp3_amd.trainer import TrainingModule
    from stp3_amd.utils import to_channels_last

    torch.manual_seed(1234)
    half = bev_cells * 0.25
    if full_losses is None:                    # bench.py's workload: depth + instance + flow losses on top
        full_losses = os.environ.get('STP3_DRYRUN_FULL_LOSSES', '1') == '1'
    extra = {'LIFT.GT_DEPTH': True, 'INSTANCE_SEG.ENABLED': True, 'INSTANCE_FLOW.ENABLED': True} if full_losses else {}
    cfg = perception_cfg(**{'IMAGE.FINAL_DIM': final_dim, 'LIFT.X_BOUND': [-half, half, 0.5],
                            'LIFT.Y_BOUND': [-half, half, 0.5], **extra})
    module = to_channels_last(TrainingModule(cfg.convert_to_dict()))
    module.train()
    batch = synthetic.make_batch(batch=batch_size, seq=3, final_dim=final_dim, bev=(bev_cells, bev_cells), seed=0,
                                 gt_depth=full_losses, instance=full_losses)
    return module, batch, cfg


def drive(recorder):
    import torch
    module, batch, _ = dry_setup(recorder)
    log = open(os.environ['STP3_TRACE_LOG'], 'a')

    def mark(text):
        log.write(f'# {text}\n')
        log.flush()

    model = module.model
    from stp3_amd.parallel import FlatAdam, GradientBuckets
    buckets = GradientBuckets(model)
    opt = FlatAdam(buckets, lr=1e-3, weight_decay=1e-7)
    mark('plan')
    model.prepare_plan(batch['intrinsics'], batch['extrinsics'], batch['future_egomotion'], torch.device('cpu'))
    for it in range(2):                       # bench.py's eager step, twice: the second one sees updated weights
        mark(f'step {it}: forward')
        buckets.zero_grad()
        with torch.autocast('cpu', dtype=torch.bfloat16):
            loss = module.training_step(batch)
        mark(f'step {it}: backward loss={tuple(loss.shape)}/{loss.dtype}')
        loss.backward()
        buckets.finish()
        missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
        mark(f'parameters without gradient: {missing}')
        shapes_ok = all(p.grad.shape == p.shape for p in model.parameters() if p.grad is not None)
        in_buckets = all(p.grad is v for (_, ps), vs in zip(buckets.buckets, buckets.grad_views) for p, v in zip(ps, vs))
        mark(f'gradient shapes ok: {shapes_ok and in_buckets}')
        mark(f'step {it}: clip + optimizer')
        opt.clip_and_step(5.0)
    mark('end')


if __name__ == '__main__':
    drive(sys.argv[1])
