"""A/B of two builds of the library inside one GPU session: bench.py with libstp3hip.so replaced by the file named in EXP_LIB
(an experiment build, e.g. `hipcc ... -o st-p3_amd/stp3_amd/libstp3hip_exp.so`).  Not used by anything else.

    EXP_LIB=/root/repo/st-p3_amd/stp3_amd/libstp3hip_exp.so python scripts/bench_with_lib.py --no-cpu-baseline
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
from stp3_amd import _lib  # noqa: E402

if os.environ.get('EXP_LIB'):
    _lib.LIB_PATH = os.environ['EXP_LIB']
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
