"""Greedy A/B of the experimental switches on the MI355X: keep a switch if the step gets faster with it.

    python scripts/ab_greedy.py [--steps 10] [--warmup 3] [--workload c3] [--min-gain 0.5] [--out gpurun_out/ab.json]

Runs bench.py (no roofline / CPU-baseline legs) once for the baseline and once per candidate on top of what has been
kept so far, in the order below (validated-on-hardware switches first, the ones with the largest expected effect
early).  A candidate that fails, produces a non-finite loss or is not at least --min-gain percent faster is dropped.
Prints one line per run and, at the end, the environment to export.  Every switch must have passed its parity tests
first (scripts/gpu_round2_validate.sh runs those): this script only measures.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CANDIDATES = [
    ('bn_geom', {'STP3_BN_GEOM': '1'}),
    ('fused_se', {'STP3_FUSED_SE': '1'}),
    ('se_mlp', {'STP3_FUSED_SE': '1', 'STP3_SE_MLP': '1'}),
    ('label_warp', {'STP3_LABEL_WARP': 'batched'}),
    ('grad_gather', {'STP3_GRAD_GATHER': '1'}),
    ('weight_prep', {'STP3_WEIGHT_PREP': '1'}),
    ('cpp_ops', {'STP3_CPP_OPS': '1'}),
    ('conv_kernel_v2', {'STP3_CONV_KERNEL': 'v2'}),
    ('conv_v2', {'STP3_CONV_V2': '1'}),
    ('fused_adam', {'STP3_FUSED_ADAM': '1'}),
    ('lazy_bn_counter', {'STP3_LAZY_BN_COUNTER': '1'}),
    ('lift_mfma', {'STP3_LIFT_FWD': 'mfma', 'STP3_LIFT_BWD': 'mfma'}),
    ('wgrad_64', {'STP3_WGRAD_MIN_CHANNELS': '64'}),
    ('wgrad_all', {'STP3_WGRAD_MIN_CHANNELS': '0'}),      # no vendor-library weight gradients at all
    ('mfma_conv_all', {'STP3_MFMA_CONV': 'all'}),
    ('miopen_find', {'STP3_MIOPEN_FIND': '1'}),           # slow first step: give it --timeout 900
]


PASS_THROUGH = ('STP3_BENCH_DRYRUN', 'STP3_HOST_DRYRUN', 'STP3_TRACE_LOG', 'STP3_REAL_LIB', 'STP3_BENCH_VERBOSE')


def run(env_extra, args):
    env = {k: v for k, v in os.environ.items() if not k.startswith('STP3_') or k in PASS_THROUGH}
    env.update(STP3_GRAD_GATHER='0', STP3_LABEL_WARP='per_label', STP3_LAZY_BN_COUNTER='0')      # bench.py turns these on unless told otherwise
    env.update(env_extra)
    cmd = [sys.executable] + (args.bench_cmd.split() if args.bench_cmd else [os.path.join(ROOT, 'bench.py')]) + [
        '--steps', str(args.steps), '--warmup', str(args.warmup), '--batch', str(args.batch), '--no-cpu-baseline',
        '--no-roofline', '--workload', args.workload]
    t0 = time.time()
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.timeout)
    except subprocess.TimeoutExpired:
        return None, f'timeout after {args.timeout} s'
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    if out.returncode != 0 or not lines:
        tail = [l for l in out.stderr.splitlines() if 'amdgpu.ids' not in l and 'MIOpen(HIP)' not in l][-2:]
        return None, f'rc={out.returncode} ' + ' | '.join(tail)[:300]
    line = json.loads(lines[-1])
    if args.workload not in ('perception',) and 'depth CE' not in line['config']['workload']:
        return None, 'fell back to the perception workload'
    return line['ms_per_step'], f'{time.time() - t0:.0f} s wall'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c3')
    ap.add_argument('--min-gain', type=float, default=0.5, help='percent')
    ap.add_argument('--timeout', type=float, default=300.0)
    ap.add_argument('--only', default='', help='comma-separated candidate names')
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--bench-cmd', default='', help=argparse.SUPPRESS)     # CPU dry run of this script's control flow
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'ab_greedy.json'))
    args = ap.parse_args()
    only = [n for n in args.only.split(',') if n]
    kept, log = {}, []
    base, note = run({}, args)
    print(f'baseline: {base} ms/step ({note})', flush=True)
    if base is None:
        sys.exit(1)
    repeat, _ = run({}, args)                                  # run-to-run noise of the baseline
    noise = abs(repeat - base) / base * 100.0 if repeat else float('nan')
    print(f'baseline again: {repeat} ms/step (noise {noise:.2f} %)', flush=True)
    best = min(base, repeat) if repeat else base
    log.append({'name': 'baseline', 'ms': base, 'repeat_ms': repeat})
    for name, env in CANDIDATES:
        if only and name not in only:
            continue
        trial = dict(kept, **env)
        ms, note = run(trial, args)
        gain = (best - ms) / best * 100.0 if ms else None
        keep = ms is not None and gain >= args.min_gain
        print(f'{name:16s} {"KEEP" if keep else "drop"}  {ms} ms/step  gain {gain if gain is None else round(gain, 2)} %  ({note})',
              flush=True)
        log.append({'name': name, 'env': env, 'ms': ms, 'gain_percent': gain, 'kept': keep, 'note': note})
        if keep:
            kept, best = trial, ms
    print('\nbest: %.2f ms/step (baseline %.2f)' % (best, base))
    print('export ' + ' '.join(f'{k}={v}' for k, v in sorted(kept.items())))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'w') as f:
        json.dump({'baseline_ms': base, 'best_ms': best, 'kept': kept, 'runs': log}, f, indent=1)


if __name__ == '__main__':
    main()
