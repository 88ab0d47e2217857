#!/bin/bash
# round 4, GPU job 18: scale-gradient traversal + finalize in ONE launch (k_batch_traverse_fin, tickets) against the separate finalize launch
# (same development library, LQ_TUNE_BATCH_TICKETS=0/1), then the batch parity file through the product library (tickets on)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job18
mkdir -p $O
CS=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc
timeout -k 10 600 python3 -m pytest tests/test_gpu_batch.py -q -m gpu -x -p no:cacheprovider > $O/pytest_batch.txt 2>&1
rc=$?
echo "batch parity (product library, tickets on) rc=$rc" | tee -a $O/status.txt
tail -3 $O/pytest_batch.txt
[ $rc -eq 0 ] || exit 1
BW="python3 tools/bench_weights.py --abi-only --kernel-storage oihw --steps 300"
for rep in 1 2 3; do
for tk in 0 1; do
  for cfg in imagenette:channelwise resnet50:channelwise; do
    LQ_HIP_LIB=$CS/liblq_hip_dev.so LQ_TUNE_BATCH_TICKETS=$tk timeout -k 10 120 $BW --only $cfg 2>>$O/sweep.err | grep '^{' | sed "s/^{/{\"tickets\": $tk, /" >> $O/sweep.jsonl || exit 1
  done
done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job18/sweep.jsonl"):
    r=json.loads(l)
    print("tickets", r["tickets"], r["config"], r["orientation"], "abi %.1f  fused %.1f" % (r["us_per_step_batched_abi"], r["us_per_step_batched_abi_oihw_fused_update"]))
PY
for tk in 0 1; do
 for cfg in imagenette:channelwise resnet50:channelwise; do
  d=$O/stats_tk${tk}_$(echo $cfg | tr ':' '_'); mkdir -p $d
  LQ_HIP_LIB=$CS/liblq_hip_dev.so LQ_TUNE_BATCH_TICKETS=$tk timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_weights.py --only $cfg --abi-only --kernel-storage oihw --steps 200 > $d/run.log 2>&1 || exit 1
  find $d -name '*kernel_trace.csv' -delete
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$tk $cfg" <<'PY'
import csv,sys
print("== tickets", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if 'lq::k_batch' in r['Name']: print("%-44s calls %s avg %.2f min %.2f max %.2f us" % (r['Name'][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
 done
done
