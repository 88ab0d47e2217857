#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_stats_k; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_stats.py > $out/run.log 2> $out/err.log
f=$(ls -S $out/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lq::k_q" in r["Name"]:
        print(f'{r["Name"][:60]:60s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.2f} min={float(r["MinNs"])/1e3:8.2f} max={float(r["MaxNs"])/1e3:8.2f}')
PY
