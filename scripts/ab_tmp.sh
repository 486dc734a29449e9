mkdir -p gpurun_out/r05q
for w in planning; do
  for l in graph; do
    timeout 600 python bench.py --no-cpu-baseline --steps 30 --workload $w --launch $l > gpurun_out/r05q/bench_${w}_$l.json 2> gpurun_out/r05q/bench_${w}_$l.err
    echo "$w $l rc=$?"; grep -v "Warning\|self.backbone\|amdgpu.ids" gpurun_out/r05q/bench_${w}_$l.err | tail -8 | cut -c1-300
    python -c "
import json; b=json.load(open('gpurun_out/r05q/bench_${w}_$l.json')); print('$w', b['config']['launch'], 'ms', b['ms_per_step'], 'samples/s', b['value'], 'host', b['host_enqueue_ms_per_step'])"
  done
done
timeout 600 python -m pytest tests/test_planning_gpu.py tests/test_prediction_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
