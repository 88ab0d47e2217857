#!/bin/bash
# development helper: how much of the step is launch/event overhead?
show() { python -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['value']), 'img/s', round(d['ms_per_step']*1000,1), 'us/step', round(d['config']['step_GBs']), 'GB/s step')"; }
for v in split fused; do
  python bench.py --no-cpu-baseline --lam 1e-11 --variant $v --steps 400 > gpurun_out/o1.json 2>>gpurun_out/bench.err && show gpurun_out/o1.json "$v events-every-step"
  python bench.py --no-cpu-baseline --lam 1e-11 --variant $v --steps 400 --event-every 0 > gpurun_out/o2.json 2>>gpurun_out/bench.err && show gpurun_out/o2.json "$v no-events"
  python bench.py --no-cpu-baseline --lam 1e-11 --variant $v --steps 400 --event-every 8 > gpurun_out/o3.json 2>>gpurun_out/bench.err && show gpurun_out/o3.json "$v events-every-8"
  python bench.py --no-cpu-baseline --lam 1e-11 --variant $v --steps 400 --event-every 0 --graph > gpurun_out/o4.json 2>>gpurun_out/bench.err && show gpurun_out/o4.json "$v graph no-events"
done
tail -3 gpurun_out/bench.err
