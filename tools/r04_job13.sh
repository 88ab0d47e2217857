#!/bin/bash
# round 4, GPU job 13: elements per block of the per-column tasks (1x1 kernels) of the scale-gradient pass: fewer row blocks = fewer partials
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job13
mkdir -p $O
CS=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc
BW="python3 tools/bench_weights.py --abi-only --kernel-storage oihw --steps 300"
for rep in 1 2; do
for m in 100 150 200 300 400; do
  for cfg in resnet50:channelwise imagenette:channelwise; do
    LQ_HIP_LIB=$CS/liblq_hip_dev_S.so LQ_TUNE_BATCH_W1MUL=$m timeout -k 10 120 $BW --only $cfg 2>>$O/sweep.err | grep '^{' | sed "s/^{/{\"lib\": \"S\", \"w1mul\": $m, /" >> $O/sweep.jsonl
  done
done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job13/sweep.jsonl"):
    r=json.loads(l)
    print(r["w1mul"], r["config"], "abi %.1f  fused %.1f" % (r["us_per_step_batched_abi"], r["us_per_step_batched_abi_oihw_fused_update"]))
PY
for m in 100 200; do
  mkdir -p $O/stats_$m
  LQ_HIP_LIB=$CS/liblq_hip_dev_S.so LQ_TUNE_BATCH_W1MUL=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- python3 tools/bench_weights.py --only resnet50:channelwise --abi-only --kernel-storage oihw --steps 200 > $O/stats_$m/run.log 2>&1
  find $O/stats_$m -name '*kernel_trace.csv' -delete
  f=$(find $O/stats_$m -name '*kernel_stats.csv' | head -1)
  python3 - "$f" $m <<'PY'
import csv,sys
print("== w1mul", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if 'lq::k_batch' in r['Name']: print("%-44s calls %s avg %.2f min %.2f max %.2f us" % (r['Name'][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
