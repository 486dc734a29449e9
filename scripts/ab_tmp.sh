mkdir -p gpurun_out/r05f
run() { # tag AB
  AB="$2" timeout 300 python scripts/bench_ab.py --no-cpu-baseline --no-roofline --steps 60 > gpurun_out/r05f/$1.json 2> gpurun_out/r05f/$1.err
  python -c "
import json; b=json.load(open('gpurun_out/r05f/$1.json')); print('$1', 'ms', b['ms_per_step'], 'host', b['host_enqueue_ms_per_step'])"
}
for rep in 1 2 3; do
  run base.$rep "stp3_amd.models.encoder.PARALLEL_HEADS=0,stp3_amd.ops.SIDE_WGRAD=0"
  run heads.$rep "stp3_amd.models.encoder.PARALLEL_HEADS=1,stp3_amd.ops.SIDE_WGRAD=0"
  run wgrad.$rep "stp3_amd.models.encoder.PARALLEL_HEADS=0,stp3_amd.ops.SIDE_WGRAD=1"
  run both.$rep "stp3_amd.models.encoder.PARALLEL_HEADS=1,stp3_amd.ops.SIDE_WGRAD=1"
done
timeout 900 python -m pytest tests/test_train_parity_gpu.py tests/test_step_parity_gpu.py tests/test_recompute_gpu.py tests/test_conv_gpu.py tests/test_fused_ops_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300
AB="stp3_amd.models.encoder.PARALLEL_HEADS=1,stp3_amd.ops.SIDE_WGRAD=1" bash scripts/gpu_r05.sh r05f profile 2>&1 | tail -50 | cut -c1-200
