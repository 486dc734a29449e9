#!/bin/bash
# GPU visit for the voxel pool: parity tests, HIP-event timings, per-kernel rocprofv3 stats.  Output: gpurun_out/$1
out=gpurun_out/${1:-lift}; mkdir -p $out
timeout 900 python -m pytest tests/test_lift_gpu.py tests/test_lift_stress_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 | tee $out/pytest.log
timeout 600 python scripts/time_lift.py 2>&1 | tee $out/time_lift.log
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lift -o lift -- python $GRAFT_REPO_ROOT/scripts/time_lift.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_lift -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
python - <<PY
import csv
rows=list(csv.DictReader(open('$out/kernel_stats.csv')))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('lift_','plan_','depth_softmax','grad_import','transpose_kernel')):
        print(f"{n[:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
