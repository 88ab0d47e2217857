#!/bin/bash
# usage: tools/prof_weights.sh [config:orientation]  -- rocprofv3 kernel stats of tools/bench_weights.py (latency regime)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_weights; mkdir -p $out
if [ -n "$1" ]; then only="--only $1"; else only=""; fi
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_weights.py $only > $out/run.log 2> $out/err.log
f=$(ls -S $out/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lq::" in r["Name"] and "selftest" not in r["Name"]:
        print(f'{r["Name"][:72]:72s} calls={r["Calls"]:>6s} avg={float(r["AverageNs"])/1e3:7.2f} min={float(r["MinNs"])/1e3:7.2f} max={float(r["MaxNs"])/1e3:7.2f}')
PY
