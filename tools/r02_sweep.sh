#!/bin/bash
# usage: tools/r02_sweep.sh <outdir>   (GPU box, repo root) -- rocprofv3 kernel durations of the round-2 streaming forms
# under their development knobs (LQ_TUNE_S2 / LQ_TUNE_PIPE / LQ_TUNE_COL_RB / LQ_TUNE_TINY_U), one process per point.
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
run() {   # run <tag> <outer> <G> <inner> [ENV=VAL ...]
  tag=$1; o=$2; g=$3; i=$4; shift 4
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 tools/shape_case.py $o $g $i --iters 20 > $out/$tag.log 2>&1 ) || { echo "FAILED $tag"; return 1; }
  python3 - "$out/$tag" "$tag" "$o" "$g" "$i" >> $out/sweep.txt <<'PY'
import csv, glob, sys
d, tag, o, g, i = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = o * g * i
f = glob.glob(f"{d}/*/*kernel_stats.csv")[0]
B = {"0": 8, "1": 8, "2": 12}
for r in csv.DictReader(open(f)):
    nm = r["Name"]
    if "lq::" not in nm or "selftest" in nm:
        continue
    k = nm.split("(")[0].replace("void lq::", "")
    op = k.split("<")[1].split(",")[0].split(">")[0] if "<" in k else ""
    us = float(r["AverageNs"]) / 1e3
    gbs = B.get(op, 0) * n / us / 1e3 if "finalize" not in k else 0
    print(f"{tag:34s} {k:44s} calls={r['Calls']:>3s} avg={us:7.1f}us min={float(r['MinNs'])/1e3:7.1f}us {gbs:6.0f} GB/s")
PY
}
C256="150528 256 1"; C6144="6144 6144 1"; IN8="2048 2048 8"; C64="602112 64 1"; C3="12845056 3 1"; R32="1 1048576 32"; R512="1 65536 512"
C16="2408448 16 1"; C32="1204224 32 1"; C8="4816896 8 1"
for rep in a b; do
  run c64_$rep $C64 || exit 1
  run c16_$rep $C16 || exit 1
  run c8_$rep $C8 || exit 1
  run c3_$rep $C3 || exit 1
  run nchw_$rep 256 3 50176 || exit 1
  run nchw_u4_$rep 256 3 50176 LQ_TUNE_U4=1 || exit 1
  run hwio_u4_$rep 9 2048 2048 LQ_TUNE_U4=1 || exit 1
  run hwio_$rep 9 2048 2048 || exit 1
done
echo "sweep done"
