"""A/B of host-side switches inside one GPU session: bench.py with module-level flags overridden from AB, e.g.
    AB="stp3_amd.ops.DIRECT_BUCKET_GRADS=0" python scripts/bench_ab.py --no-cpu-baseline
(and EXP_LIB=<path> for another build of the library).  Prints bench.py's line; not used by anything else."""
import importlib
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
from stp3_amd import _lib  # noqa: E402

if os.environ.get('EXP_LIB'):
    _lib.LIB_PATH = os.environ['EXP_LIB']
for item in filter(None, os.environ.get('AB', '').split(',')):
    target, value = item.split('=')
    module, attr = target.rsplit('.', 1)
    setattr(importlib.import_module(module), attr, type(getattr(importlib.import_module(module), attr))(int(value)))
    print(f'[ab] {target} = {getattr(importlib.import_module(module), attr)!r}', file=sys.stderr)
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
