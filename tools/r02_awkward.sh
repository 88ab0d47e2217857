#!/bin/bash
# usage: tools/r02_awkward.sh <outdir> [ENV=VAL ...]   (GPU box, repo root)
# rocprofv3 kernel durations of K1 / K2 / K4 on descriptors OFF the friendly grid (odd row lengths, odd column counts,
# small odd inner extents), 25-40 M elements each: where does a traversal fall off its streaming form?
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
for kv in "$@"; do export "$kv"; done
run() {   # run <tag> <outer> <G> <inner>
  tag=$1; o=$2; g=$3; i=$4
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 tools/shape_case.py $o $g $i --iters 12 > $out/$tag.log 2>&1 || { echo "FAILED $tag"; return 1; }
  python3 - "$out/$tag" "$tag" "$o" "$g" "$i" >> $out/sweep.txt <<'PY'
import csv, glob, sys
d, tag, o, g, i = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = o * g * i
f = glob.glob(f"{d}/*/*kernel_stats.csv")[0]
B = {"0": 8, "1": 8, "2": 12}
row = {}
for r in csv.DictReader(open(f)):
    nm = r["Name"]
    if "lq::" not in nm or "selftest" in nm or "finalize" in nm:
        continue
    k = nm.split("(")[0].replace("void lq::", "")
    op = k.split("<")[1].split(",")[0].split(">")[0] if "<" in k else ""
    us = float(r["AverageNs"]) / 1e3
    if op in B:
        row[op] = (k, us, B[op] * n / us / 1e3)
print(f"{tag:18s} ({o},{g},{i})".ljust(44) + " | ".join(f"{row[op][0]:38s} {row[op][1]:7.1f}us {row[op][2]:5.0f}" if op in row else "-" for op in ("0", "1", "2")))
PY
}
if [ -n "$CASES" ]; then           # CASES="tag:outer,G,inner ..." overrides the lists below
for c in $CASES; do tag=${c%%:*}; d=${c##*:}; IFS=, read o g i <<< "$d"; run $tag $o $g $i || exit 1; done
elif [ "$MIDROWS" = "1" ]; then      # rows of 68..1020 elements with L % 4 == 0 only (A/B of LQ_TUNE_WIN)
run r100     1 327680 100  && run r300   1 114688 300  && run r512    1 65536 512  && run r1000  1 32768 1000 && run o4r196 64 2048 196 || exit 1
else
run a49      256 2048 49   && run r17    1 2097152 17  && run r100    1 327680 100 && run r300   1 114688 300 && run r1000  1 32768 1000 && \
run r1001    1 32768 1001  && run r1023  1 32768 1023  && run r1025   1 32768 1025 && run r2047  1 16384 2047 && \
run r3000    1 12288 3000  && run r50177 256 3 50177   && run o8r777  8 6144 777   && \
run c5       8388608 5 1   && run c10    3355443 10 1  && run c30     1118481 30 1 && run c100   327680 100 1 && \
run c130     262144 130 1  && run c1000  32768 1000 1  && run c1001   32768 1001 1 && run c4099  8192 4099 1 && \
run i2       4096 4096 2   && run i3     4096 2730 3   && run i5      2560 2560 5  && run i12    1664 1664 12 && run i15 1472 1472 15 || exit 1
fi
cat $out/sweep.txt
