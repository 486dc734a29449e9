#!/bin/bash
# One GPU-box visit: parity tests, lift kernel timings, then (optionally) the profile.
# Usage (from repo root on the GPU box): bash scripts/gpu_check.sh <tag> [pytest args...]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -25 $OUT/pytest.log
timeout 300 python scripts/time_lift.py > $OUT/time_lift.log 2>&1; cat $OUT/time_lift.log
