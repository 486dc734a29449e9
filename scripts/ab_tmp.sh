mkdir -p gpurun_out/r05g
timeout 600 python -m pytest tests/test_graph_step_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25 | cut -c1-400
run() { # tag args
  tag=$1; shift
  timeout 400 python bench.py --no-cpu-baseline --no-roofline --steps 60 "$@" > gpurun_out/r05g/$tag.json 2> gpurun_out/r05g/$tag.err
  echo "rc=$?"; tail -2 gpurun_out/r05g/$tag.err | cut -c1-300
  python -c "
import json; b=json.load(open('gpurun_out/r05g/$tag.json')); print('$tag', b['config']['launch'], 'ms', b['ms_per_step'], 'host', b['host_enqueue_ms_per_step'])"
}
for rep in 1 2; do
  run eager.$rep --launch eager
  run graph.$rep --launch graph
done
