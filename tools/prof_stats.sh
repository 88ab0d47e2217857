#!/bin/bash
# usage: tools/prof_stats.sh <outdir> <bench args...>   -- rocprofv3 kernel-trace + stats of bench.py, prints the lq:: kernels
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 "$@" > $out/bench.json 2> $out/err.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lq::" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.2f} min_us={float(r["MinNs"])/1e3:8.2f} max_us={float(r["MaxNs"])/1e3:8.2f}')
PY
