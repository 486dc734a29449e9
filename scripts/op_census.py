"""Operator census of one training step (no GPU needed): how many torch operators and how many C-ABI launches does
the host issue per step?  Dry run (tests/model_trace.py); the COUNT of operators does
not depend on tensor sizes, so a tiny configuration is used.

    python scripts/op_census.py
"""
import collections
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))


def worker(recorder, log):
    import torch
    from torch.profiler import ProfilerActivity, profile
    from tests import model_trace
    module, batch, cfg = model_trace.dry_setup(recorder, final_dim=(32, 48), batch_size=1, bev_cells=32,
                                               deterministic_fill=False)
    from stp3_amd.parallel import FlatAdam, GradientBuckets
    buckets = GradientBuckets(module.model)
    opt = FlatAdam(buckets, lr=1e-3, weight_decay=1e-7)
    model = module.model
    poses = (batch['intrinsics'], batch['extrinsics'], batch['future_egomotion'])
    plan = model.prepare_plan(*poses, torch.device('cpu'))

    def step():
        model.prepare_plan(*poses, torch.device('cpu'), out=plan)
        buckets.zero_grad()
        with torch.autocast('cpu', dtype=torch.bfloat16):
            loss = module.training_step(batch)
        loss.backward()
        buckets.finish()
        opt.clip_and_step(5.0)

    step()
    step()
    open(log, 'w').close()
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        step()
    ev = prof.events()
    top = collections.Counter(e.name for e in ev if e.cpu_parent is None)
    host_only = {'select', 'view', 'slice', 'reshape', 'permute', 'transpose', 'as_strided', 'detach', 'alias', 'expand',
                 'unsqueeze', 'squeeze', 't', 'flatten', 'empty', 'empty_like', 'empty_strided', 'narrow', 'unflatten',
                 '_unsafe_view', 'view_as', 'expand_as', 'contiguous', 'to', 'resize_', 'unbind', 'split', 'chunk',
                 'lift_fresh', 'detach_', 'set_', 'result_type', 'item', '_local_scalar_dense', 'is_nonzero', 'size',
                 'stride', 'numel', 'dim', 'type_as', 'movedim', 'unfold', 'split_with_sizes', 'squeeze_', 'unsqueeze_',
                 'new_empty', 'new_empty_strided', 'conj', '_reshape_alias', 'diagonal', 'real', 'broadcast_to'}
    leaf = [e.name[6:] for e in ev if e.name.startswith('aten::') and not any(
        c.name.startswith('aten::') for c in e.cpu_children)]
    kernels = collections.Counter(n for n in leaf if n not in host_only)
    aten_leaf = sum(kernels.values())
    calls = collections.Counter(l.split(' ', 1)[0] for l in open(log) if l.startswith('stp3_')
                                and not l.split(' ', 1)[0].endswith(('_bytes', '_workspace')))
    print(f'top-level operators per step: {sum(top.values())}   torch operators that launch kernels: {aten_leaf}   '
          f'C-ABI launches: {sum(calls.values())}')
    print('  largest top-level groups: ' + ', '.join(f'{k.split(": ")[-1]} x{v}' for k, v in top.most_common(8)))
    print('  torch kernels: ' + ', '.join(f'{k} x{v}' for k, v in kernels.most_common(int(os.environ.get("CENSUS_TOP", "14")))))
    print('  C-ABI: ' + ', '.join(f'{k[5:]} x{v}' for k, v in calls.most_common()))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        worker(sys.argv[2], sys.argv[3])
        return
    from tests import host_trace
    with tempfile.TemporaryDirectory() as tmp:
        rec = host_trace.build_recorder(os.path.join(tmp, 'librec.so'))
        log = os.path.join(tmp, 'trace.log')
        env = dict(os.environ, STP3_HOST_DRYRUN='1', STP3_TRACE_LOG=log,
                   STP3_REAL_LIB=os.path.join(ROOT, 'st-p3_amd', 'stp3_amd', 'libstp3hip.so'))
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--worker', rec, log], env=env,
                              stderr=subprocess.DEVNULL)


if __name__ == '__main__':
    main()
