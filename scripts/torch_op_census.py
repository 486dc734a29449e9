"""Which torch operators are still launched inside one training step, and by which line of the package?  (no GPU needed)

Dry run of bench.py's eager step (tests/model_trace.py) under a TorchDispatchMode: every aten operator that
would be a GPU kernel of its own (copies, adds, casts, fills ...) is counted per innermost stp3_amd source line (or
"autograd engine" when none is on the stack).  The operator count of the step does not depend on the image size.

    python scripts/torch_op_census.py [--top 40] [--depth 3]
"""
import argparse
import collections
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))

KERNEL_OPS = ('aten::copy_', 'aten::add_', 'aten::add', 'aten::mul', 'aten::mul_', 'aten::fill_', 'aten::zero_', 'aten::cat',
              'aten::sum', 'aten::div', 'aten::div_', 'aten::sub', 'aten::clone', 'aten::index_select', 'aten::mean',
              'aten::_to_copy', 'aten::sigmoid', 'aten::neg', 'aten::exp', 'aten::where', 'aten::clamp', 'aten::sqrt',
              'aten::bmm', 'aten::mm', 'aten::matmul', 'aten::linalg_inv', 'aten::stack', 'aten::index', 'aten::gather',
              'aten::masked_fill_', 'aten::rsub', 'aten::pow', 'aten::abs', 'aten::max', 'aten::min', 'aten::argmax')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--depth', type=int, default=1, help='frames of the package per site')
    ap.add_argument('--worker', default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.worker is None:
        from tests import host_trace
        with tempfile.TemporaryDirectory() as tmp:
            rec = host_trace.build_recorder(os.path.join(tmp, 'librec.so'))
            env = dict(os.environ, STP3_HOST_DRYRUN='1', STP3_TRACE_LOG=os.devnull,
                       STP3_REAL_LIB=os.path.join(ROOT, 'st-p3_amd', 'stp3_amd', 'libstp3hip.so'))
            subprocess.check_call([sys.executable, os.path.abspath(__file__), '--worker', rec, '--top', str(args.top), '--depth', str(args.depth)], env=env)
        return

    import torch
    from tests import model_trace
    module, batch, cfg = model_trace.dry_setup(args.worker, final_dim=(64, 96), batch_size=2, bev_cells=64, deterministic_fill=False)
    from stp3_amd.parallel import FlatAdam, GradientBuckets
    buckets = GradientBuckets(module.model)
    opt = FlatAdam(buckets, lr=cfg.OPTIMIZER.LR, weight_decay=cfg.OPTIMIZER.WEIGHT_DECAY)
    model = module.model
    # as the captured step runs it (stp3_amd/graph.py): poses, label-warp matrices and the plan prepared ahead of the step
    prepared = module.prepare_batch(batch, torch.device('cpu'))
    gbatch = dict(batch, future_egomotion=prepared['ego'], _prepared=prepared)

    def step():
        buckets.zero_grad()
        model.prebuilt_plan = prepared['plan']
        with torch.autocast('cpu', dtype=torch.bfloat16):
            loss = module.training_step(gbatch)
        model.prebuilt_plan = None
        loss.backward()
        buckets.finish()
        opt.clip_and_step(5.0)

    for _ in range(2):
        step()

    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    by_site = collections.Counter()
    by_op = collections.Counter()
    views = ('view', 'reshape', 'slice', 'select', 'expand', 'permute', 'transpose', 'as_strided', 'detach', 'alias', 'unsqueeze',
             'squeeze', 'split', 'unbind', 't.default', 'empty', 'narrow', 'chunk', 'unfold', 'size', 'stride', 'is_', 'sym_', '_local_scalar',
             'lift_fresh', 'item', 'new_empty', 'resize', 'set_', 'storage_offset', 'numel', 'dim', 'contiguous', 'prim.', '_unsafe_view',
             'diagonal', 'real', 'result_type', 'can_cast', '_has_compatible', 'is_pinned', 'is_same_size', 'equal')

    class Census(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None, depth=args.depth):
            name = str(func)
            if not any(v in name for v in views):
                site = 'autograd engine / torch'
                chain = [f"{fr.filename.split('stp3_amd/')[-1]}:{fr.lineno} {fr.name}" for fr in reversed(traceback.extract_stack()[:-1])
                         if 'stp3_amd/' in fr.filename and 'site-packages' not in fr.filename]
                if chain:
                    site = ' < '.join(chain[:depth])
                shape = ''
                for a in args:
                    if isinstance(a, torch.Tensor):
                        shape = f'{tuple(a.shape)} {str(a.dtype)[6:]}'
                        break
                by_site[(name, site)] += 1
                by_op[name] += 1
            return func(*args, **(kwargs or {}))

    with Census():
        step()
    print('top-level operator launches in one step:', sum(by_op.values()))
    for op, n in by_op.most_common():
        print(f'  {n:5d}  {op}')
    print()
    for (op, site), n in by_site.most_common(args.top):
        print(f'  {n:5d}  {op:18s} {site}')


if __name__ == '__main__':
    main()
