#!/bin/bash
# One GPU visit: the whole -m gpu suite, the convolution timings, one bench.py line.  Output under gpurun_out/$1.
out=gpurun_out/${1:-visit}; mkdir -p $out
export STP3_PARITY_REPORT=$out/parity.json
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -40 $out/pytest.log | grep -v "^\[parity\]" | tail -25
timeout 600 python scripts/time_conv.py > $out/time_conv.log 2>&1; tail -40 $out/time_conv.log
timeout 300 python scripts/time_small_ops.py > $out/time_small_ops.log 2>&1; cat $out/time_small_ops.log
timeout 900 python bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cat $out/bench.json
