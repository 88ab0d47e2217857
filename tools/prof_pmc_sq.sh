#!/bin/bash
# usage: tools/prof_pmc_sq.sh <outdir> <bench args...> -- SQ stall/issue counters of the lq:: kernels (one pass, 6 SQ slots)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 8 --event-every 0 "$@" > $out/bench.json 2> $out/err.log
python3 - $out <<'PY'
import csv,glob,sys,collections,json
out=sys.argv[1]
f=sorted(glob.glob(f"{out}/*/*counter_collection.csv"), key=lambda p: -__import__("os").path.getsize(p))[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "lq::k_row_stream" in r["Kernel_Name"] or "lq::k_finalize" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res={}
for k,v in agg.items():
    m={c: sum(x)/len(x) for c,x in v.items()}
    wc=m.get("SQ_WAVE_CYCLES",0) or 1
    m["frac_wait_any"]=m.get("SQ_WAIT_ANY",0)/wc
    m["frac_wait_inst_any"]=m.get("SQ_WAIT_INST_ANY",0)/wc
    m["frac_active_inst_any"]=m.get("SQ_ACTIVE_INST_ANY",0)/wc
    m["frac_active_valu"]=m.get("SQ_ACTIVE_INST_VALU",0)/wc
    res[k]=m
    print(k[:60], {a: (round(b,3) if b<10 else round(b)) for a,b in m.items()})
json.dump(res, open(f"{out}/sq_summary.json","w"), indent=1)
PY
