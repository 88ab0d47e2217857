#!/bin/bash
# round 4, GPU job 19: where the time of k_batch_traverse_fin goes: ablations (16 = traversal alone, 32 = tickets without the finalize), register budget (wpe0 = no waves-per-SIMD attribute)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job19
mkdir -p $O
CS=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc
for lib in dev; do
 for ab in 0 16 32; do
  d=$O/stats_${lib}_ab$ab; mkdir -p $d
  LQ_HIP_LIB=$CS/liblq_hip_$lib.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_weights.py --only imagenette:channelwise --abi-only --kernel-storage oihw --steps 200 --ablate $ab > $d/run.log 2>&1 || exit 1
  find $d -name '*kernel_trace.csv' -delete
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$lib ablate $ab" <<'PY'
import csv,sys
print("==", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if 'lq::k_batch' in r['Name']: print("%-44s calls %s avg %.2f min %.2f max %.2f us" % (r['Name'][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
 done
done
