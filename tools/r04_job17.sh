#!/bin/bash
# round 4, GPU job 17: re-entry check on a rebuilt library (the container was re-created): the driver's GPU command, smoke, default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job17
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu_full_suite.txt 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -2 $O/pytest_gpu_full_suite.txt
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/status.txt
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench.err
echo "bench rc=$?" | tee -a $O/status.txt
python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); e=d['extras']; print({k:(v.get('us_per_step'), v.get('frac_of_8TBs')) for k,v in e.items() if k.startswith('weight_set')})"
