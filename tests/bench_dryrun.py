"""CPU test infrastructure: run bench.py's main() as a dry run (see tests/model_trace.py).

    STP3_BENCH_DRYRUN=1 STP3_TRACE_LOG=... STP3_REAL_LIB=... python tests/bench_dryrun.py recorder.so [bench args]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == '__main__':
    from tests import model_trace
    model_trace.patch_process(sys.argv[1], deterministic_fill=False)
    import bench
    sys.argv = ['bench.py'] + sys.argv[2:]
    bench.main()
