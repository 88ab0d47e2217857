#!/bin/bash
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
run() { tag=$1; o=$2; g=$3; i=$4; shift 4
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 tools/shape_case.py $o $g $i --iters 20 --ops k1 > $out/$tag.log 2>&1 ) || { echo "FAILED $tag"; return 1; }
  python3 - "$out/$tag" "$tag" "$o" "$g" "$i" >> $out/sweep.txt <<'PY'
import csv, glob, sys
d, tag, o, g, i = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = o * g * i
f = glob.glob(f"{d}/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    nm = r["Name"]
    if "lq::" not in nm or "selftest" in nm: continue
    k = nm.split("(")[0].replace("void lq::", "")
    us = float(r["AverageNs"]) / 1e3
    print(f"{tag:24s} {k:44s} calls={r['Calls']:>3s} avg={us:7.1f}us min={float(r['MinNs'])/1e3:7.1f}us {8*n/us/1e3:6.0f} GB/s")
PY
}
for rep in a b; do
run u4100_new_$rep 1 8192 4100 && run u4100_old_$rep 1 8192 4100 LQ_TUNE_S2=256 && run u4099_new_$rep 1 8192 4099 && run u4099_old_$rep 1 8192 4099 LQ_TUNE_S2=256 && run u1028_new_$rep 1 32768 1028 && run u1028_old_$rep 1 32768 1028 LQ_TUNE_S2=256 && run u1001_new_$rep 1 32768 1001 && run u1001_old_$rep 1 32768 1001 LQ_TUNE_S2=256 || exit 1
done
cat $out/sweep.txt
