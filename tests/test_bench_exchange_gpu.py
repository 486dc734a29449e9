"""GPU: ``bench.py`` on the N > 1 code path of the training step (reference recipe: DDP + sync_batchnorm,
/root/reference/train.py:43-56) -- ``--force-exchange`` runs it in a process group of ONE RCCL rank (RCCL refuses two ranks on one
device): the step is captured into a hipGraph WITH its collectives, the line says so, counts the collectives of a replayed
step (counted at the capture) and reports a finite rate.  tests/test_graph_exchange_gpu.py pins the numerics; this pins the
script path the multi-GPU bench takes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_captures_the_exchange_step():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--force-exchange', '--batch', '1', '--steps', '3', '--warmup', '2',
                          '--no-cpu-baseline', '--no-roofline'], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert out.returncode == 0 and lines, out.stderr[-1500:]
    d = json.loads(lines[0])
    assert d['config']['launch'] == 'hipgraph', d['config']['launch']
    assert 'force-exchange' in d['config']['parallelism'] and d['n_gpus'] == 1
    c = d['collectives_per_step']
    assert c['batchnorm_statistics_all_reduces'] > 100 and c['gradient_bucket_all_reduces'] >= 1, c
    assert d['value'] > 0 and d['ms_per_step'] > 0
