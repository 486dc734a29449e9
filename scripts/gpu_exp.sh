#!/bin/bash
# experiment: per-kernel times of library variants under exp_variants/
export TMPDIR=/tmp
for lib in exp_variants/*.so; do
  name=$(basename $lib .so)
  rm -rf /tmp/prof_$name
  (cd /tmp && EXP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o x -- python $GRAFT_REPO_ROOT/scripts/time_lift.py > /tmp/$name.log 2>&1)
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  echo "== $name"; grep -E "fwd|Error|error" /tmp/$name.log | head -3
  python - <<PY
import csv
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if any(k in n for k in ('lift_column','lift_gather','plan_')):
        print(f"   {n[:60]:60s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
done
