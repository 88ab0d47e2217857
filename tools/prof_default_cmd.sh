#!/bin/bash
# rocprofv3 --kernel-trace --stats of EXACTLY the default bench command (python3 bench.py), lq:: kernels only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_default_cmd
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py > $out/bench.json 2> $out/err.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lq::" in r["Name"]:
        print(f'{r["Name"][:74]:74s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.2f}')
PY
tail -c 600 $out/bench.json
