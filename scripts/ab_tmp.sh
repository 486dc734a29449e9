mkdir -p gpurun_out/r05e
for rep in 1 2; do
for v in 0 1; do
  AB="stp3_amd.models.encoder.PARALLEL_HEADS=$v" timeout 300 python scripts/bench_ab.py --no-cpu-baseline --no-roofline --steps 60 > gpurun_out/r05e/heads$v.$rep.json 2> gpurun_out/r05e/heads$v.$rep.err
  python -c "
import json; b=json.load(open('gpurun_out/r05e/heads$v.$rep.json')); print('PARALLEL_HEADS=$v rep $rep ms', b['ms_per_step'], 'host', b['host_enqueue_ms_per_step'])"
done; done
timeout 600 python -m pytest tests/test_train_parity_gpu.py tests/test_step_parity_gpu.py tests/test_recompute_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300
