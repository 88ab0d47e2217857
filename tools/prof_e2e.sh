#!/bin/bash
# usage: tools/prof_e2e.sh <config> <batch>  -- rocprofv3 kernel stats of the end-to-end harness step (top kernels by time)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_e2e_$1
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/e2e_knobs.py $1 $2 0 0 > $out/run.log 2> $out/err.log
f=$(ls -S $out/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={100*float(r["TotalDurationNs"])/tot:5.1f}')
lq=sum(float(r["TotalDurationNs"]) for r in rows if "lq::" in r["Name"])
print("lq:: share of GPU time: %.2f%%" % (100*lq/tot))
PY
cat $out/run.log
