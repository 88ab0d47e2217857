#!/bin/bash
# usage: tools/r03_ablate.sh <outdir> <config:orientation> [masks...]  -- development library: kernel durations with parts of the tile kernels off
out=$1; only=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LQ_HIP_LIB=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc/liblq_hip_dev.so
for m in "$@"; do
  d=$out/m$m; mkdir -p $d
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_weights.py --only $only --abi-only --ablate $m > $d/run.log 2> $d/err.log || { tail -5 $d/err.log; exit 1; }
  f=$(ls -S $d/*/*_kernel_stats.csv | head -1)
  echo "mask $m: $(cat $d/run.log | cut -c1-200)"
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_batch_traverse" in r["Name"]:
        print(f'  {r["Name"].split("(")[0].replace("void ","")[:40]:40s} avg={float(r["AverageNs"])/1e3:7.2f} min={float(r["MinNs"])/1e3:7.2f}')
PY
done
