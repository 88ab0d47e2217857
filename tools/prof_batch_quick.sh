#!/bin/bash
# usage: tools/prof_batch_quick.sh <outdir> <config:orientation>... [-- bench_weights args]   -- rocprofv3 kernel stats only (no counters) of the raw batch ABI
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sets=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; extra=("$@"); break; fi; sets+=("$1"); shift; done
for only in "${sets[@]}"; do
  d=$out/$(echo $only | tr ':' '_'); mkdir -p $d
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_weights.py --only $only --abi-only "${extra[@]}" > $d/run.log 2> $d/err.log || exit 1
  f=$(ls -S $d/*/*_kernel_stats.csv | head -1)
  cat $d/run.log
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lq::" in r["Name"] and "selftest" not in r["Name"]:
        print(f'  {r["Name"].split("(")[0].replace("void ","")[:60]:60s} calls={r["Calls"]:>5s} avg={float(r["AverageNs"])/1e3:7.2f} min={float(r["MinNs"])/1e3:7.2f} max={float(r["MaxNs"])/1e3:7.2f}')
PY
done
