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
    if os.environ.get('STP3_TEST_BREAK_GATHER') == '1':          # fault injection for the fallback-ladder test
        from stp3_amd import parallel
        _gather = parallel.GradientBuckets._gather

        def broken(self, i):
            raise RuntimeError('injected failure in gather mode')

        parallel.GradientBuckets._gather = broken
    bench.ENTRY = [os.path.abspath(__file__), sys.argv[1]]
    sys.argv = ['bench.py'] + sys.argv[2:]
    bench.main()
