#!/bin/bash
# helper used during development on the GPU box: bench sweep over lambda for split/fused variants
mkdir -p gpurun_out
for lam in 1e-3 1e-11; do
  timeout -k 10 100 python bench.py --no-cpu-baseline --lam $lam > gpurun_out/bench_split_$lam.json 2>> gpurun_out/bench.err
  python -c "import json,sys; d=json.load(open('gpurun_out/bench_split_$lam.json')); r=d['roofline']; print('split lam=$lam', round(d['value']), 'img/s', round(d['ms_per_step']*1000,1), 'us/step fwd', round(r['t_fwd_us'],1), 'bwd', round(r['t_bwd_us'],1))"
  timeout -k 10 100 python bench.py --no-cpu-baseline --variant fused --lam $lam > gpurun_out/bench_fused_$lam.json 2>> gpurun_out/bench.err
  python -c "import json,sys; d=json.load(open('gpurun_out/bench_fused_$lam.json')); r=d['roofline']; print('fused lam=$lam', round(d['value']), 'img/s', round(d['ms_per_step']*1000,1), 'us/step', round(r['achieved']), 'GB/s')"
done
