#!/bin/bash
# round 6, visit a: baseline bench at the driver's flags, where the torch launches of a step come from, BN micro timings
out=gpurun_out/r06a; mkdir -p $out
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cat $out/bench.json
timeout 600 python scripts/small_kernel_sources.py --top 150 --aten-only > $out/sources_aten.txt 2>&1; tail -5 $out/sources_aten.txt
timeout 600 python scripts/small_kernel_sources.py --top 150 --by-time > $out/sources_time.txt 2>&1
timeout 300 python scripts/time_bn.py > $out/time_bn.txt 2>&1; cat $out/time_bn.txt
