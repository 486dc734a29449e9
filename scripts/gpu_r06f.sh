#!/bin/bash
# round 6, visit f: the whole -m gpu suite, the default bench line (driver flags), fresh lift counters
out=gpurun_out/r06f; mkdir -p $out
export STP3_PARITY_REPORT=$out/parity.json
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -12 $out/pytest.log | cut -c1-300
timeout 1200 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cut -c1-400 $out/bench.json
bash scripts/gpu_pmc_lift.sh r06f > $out/pmc_lift.log 2>&1; tail -5 $out/pmc_lift.log | cut -c1-300
