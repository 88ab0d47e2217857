#!/bin/bash
# usage: tools/prof_pmc.sh <outdir> [bench args...] -- HBM traffic of the BENCH kernels: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 passes (TCC slot limits) with --kernel-trace only; bench.py stands directly behind `--`.  Its roofline leg launches
# K1, K2 and K4, so one command covers all three.  Writes <outdir>/traffic.json in the form bench.py reads from
# profiles/traffic.json: {"csrc_sha": hash of the kernel sources, "hbm_bytes_per_launch": {"K1": .., "K2": .., "K4": ..}}.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $out/$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 6 "$@" > $out/$c/bench.json 2> $out/$c/err.log || exit 1
done
python3 - $out "$@" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(f"{out}/{c}/*/*counter_collection.csv"), key=os.path.getsize)[-1]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if ("lq::k_row_stream" in r["Kernel_Name"] or "lq::k_flat_fwd" in r["Kernel_Name"]) and r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k][c] = sum(v) / len(v)
        res[k]["launches_" + c] = len(v)
summary = {"csrc_sha": bench.csrc_sha(), "collected_with": "tools/prof_pmc.sh " + " ".join(sys.argv[2:]),
           "note": "counters in KiB; gfx950 FETCH_SIZE reports half of wide coalesced reads (MI355X_MICROARCH.md, HBM): doubled",
           "kernels": {}, "hbm_bytes_per_launch": {}}
for k, v in res.items():
    rd, wr = v.get("FETCH_SIZE", 0) * 1024 * 2, v.get("WRITE_SIZE", 0) * 1024
    summary["kernels"][k] = {"hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                             "launches": v.get("launches_FETCH_SIZE")}
    op = k.split("<")[1].split(",")[0]
    key = {"0": "K1", "1": "K2", "2": "K4"}.get(op)
    if key:
        summary["hbm_bytes_per_launch"][key] = rd + wr
    print(k, round(rd), round(wr))
json.dump(summary, open(f"{out}/traffic.json", "w"), indent=1)
PY
