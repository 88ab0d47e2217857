#!/bin/bash
# usage: tools/prof_train.sh <outdir> <train args...> -- kernel-trace stats of the end-to-end harness
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 -m learned_quantization_amd.train "$@" > $out/out.json 2> $out/err.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
lq=sum(float(r["TotalDurationNs"]) for r in rows if "lq::" in r["Name"])
calls=sum(int(r["Calls"]) for r in rows); lqc=sum(int(r["Calls"]) for r in rows if "lq::" in r["Name"])
print(f"total kernel time {tot/1e6:.2f} ms over {calls} launches; lq:: kernels {lq/1e6:.2f} ms ({100*lq/tot:.1f} %) over {lqc} launches")
for r in rows[:16]:
    print(f'{r["Name"][:100]:100s} calls={r["Calls"]:>6s} tot_ms={float(r["TotalDurationNs"])/1e6:8.2f} avg_us={float(r["AverageNs"])/1e3:8.2f}')
PY
tail -1 $out/out.json
