mkdir -p gpurun_out/r05o
PREV=$PWD/st-p3_amd/exp/libstp3hip_prev.so
for rep in 1 2; do
  EXP_LIB=$PREV timeout 300 python scripts/time_conv.py > gpurun_out/r05o/time_conv_prev.$rep.log 2>&1
  timeout 300 python scripts/time_conv.py > gpurun_out/r05o/time_conv_dma.$rep.log 2>&1
done
paste -d'|' <(cut -c1-44,46-62,150-175 gpurun_out/r05o/time_conv_prev.1.log) <(cut -c46-62,150-175 gpurun_out/r05o/time_conv_dma.1.log) <(cut -c46-62,150-175 gpurun_out/r05o/time_conv_prev.2.log) <(cut -c46-62,150-175 gpurun_out/r05o/time_conv_dma.2.log) | grep -v amdgpu
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_fused_ops_gpu.py tests/test_train_parity_gpu.py tests/test_graph_step_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | cut -c1-200
for rep in 1 2; do
  for v in prev new; do
    [ $v == prev ] && export EXP_LIB=$PREV || unset EXP_LIB
    timeout 400 python scripts/bench_ab.py --no-cpu-baseline --steps 100 > gpurun_out/r05o/bench_$v.$rep.json 2> gpurun_out/r05o/bench_$v.$rep.err
    python -c "
import json; b=json.load(open('gpurun_out/r05o/bench_$v.$rep.json')); f=b['roofline_families']; print('$v $rep', b['config']['launch'], 'ms', b['ms_per_step'], 'conv', f['conv']['ms_per_step'], f['conv']['split_ms'], 'frac', f['conv']['frac'])"
  done
done
unset EXP_LIB
