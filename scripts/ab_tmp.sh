mkdir -p gpurun_out/r05h
run() { # tag AB args
  tag=$1; ab=$2; shift; shift
  AB="$ab" timeout 400 python scripts/bench_ab.py --no-cpu-baseline --no-roofline --steps 100 "$@" > gpurun_out/r05h/$tag.json 2> gpurun_out/r05h/$tag.err
  python -c "
import json; b=json.load(open('gpurun_out/r05h/$tag.json')); print('$tag', b['config']['launch'], 'ms', b['ms_per_step'], 'host', b['host_enqueue_ms_per_step'])"
}
OFF="stp3_amd.models.encoder.PARALLEL_HEADS=0,stp3_amd.ops.SIDE_WGRAD=0"
ON="stp3_amd.models.encoder.PARALLEL_HEADS=1,stp3_amd.ops.SIDE_WGRAD=1"
for rep in 1 2; do
  run eager_off.$rep "$OFF" --launch eager
  run eager_on.$rep "$ON" --launch eager
  run graph_off.$rep "$OFF" --launch graph
  run graph_on.$rep "$ON" --launch graph
done
export TMPDIR=/tmp
for v in off on; do
  [ $v == off ] && AB="$OFF" || AB="$ON"
  AB="$AB" timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g$v -o g -- python scripts/bench_ab.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --launch graph > gpurun_out/r05h/prof_$v.log 2>&1
  KT=$(find /tmp/prof_g$v -name '*kernel_trace.csv' | head -1)
  MS=$(grep '^{' gpurun_out/r05h/prof_$v.log | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "graph $v under profiler: ms_per_step $MS"
  python scripts/trace_overlap.py $KT $(python -c "print($MS*3)")
  python scripts/agg_trace.py $KT $(python -c "print($MS*3)") 12 | cut -c1-150
  rm -rf /tmp/prof_g$v
done
