#!/bin/bash
# round 4, GPU job 4: the pipelined fragment tiles -- parity (batch / layout / ddp files with the shipped library), variants x block sizes,
# per-kernel durations and the block timeline of the shipped geometry
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job4
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_layout.py tests/test_gpu_harness.py -q -m gpu > $O/pytest_subset.txt 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_subset.txt
CS=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc
BW="python3 tools/bench_weights.py --abi-only --kernel-storage oihw --steps 300"
for lib in P1 P2 P3 N; do
  for cfg in imagenette resnet50; do
    for nb in 1024 1280 1536 2048; do
      LQ_HIP_LIB=$CS/liblq_hip_dev_$lib.so LQ_TUNE_BATCH_NB=$nb timeout -k 10 120 $BW --only $cfg:channelwise 2>>$O/sweep.err | grep '^{' | sed "s/^{/{\"lib\": \"$lib\", \"nb\": $nb, /" >> $O/sweep.jsonl
    done
  done
  echo "lib $lib done" | tee -a $O/status.txt
done
for w in 4096 6144 12288 16384; do
  for cfg in imagenette resnet50; do
    LQ_HIP_LIB=$CS/liblq_hip_dev_P1.so LQ_TUNE_BATCH_W=$w timeout -k 10 120 $BW --only $cfg:channelwise 2>>$O/sweep.err | grep '^{' | sed "s/^{/{\"lib\": \"P1\", \"w\": $w, /" >> $O/sweep.jsonl
  done
done
for ab in 1 8; do
  LQ_HIP_LIB=$CS/liblq_hip_dev_P1.so timeout -k 10 120 $BW --only imagenette:channelwise --ablate $ab 2>>$O/sweep.err | grep '^{' | sed "s/^{/{\"lib\": \"P1\", \"ablate\": $ab, /" >> $O/sweep.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job4/sweep.jsonl"):
    r=json.loads(l)
    print({k:r[k] for k in ("lib","ablate","nb","w") if k in r}, r["config"], "abi %.1f  fused %.1f" % (r["us_per_step_batched_abi"], r["us_per_step_batched_abi_oihw_fused_update"]))
PY
for cfg in imagenette resnet50; do
  mkdir -p $O/stats_$cfg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$cfg -- python3 tools/bench_weights.py --only $cfg:channelwise --abi-only --kernel-storage oihw --steps 200 > $O/stats_$cfg/run.log 2>&1
  find $O/stats_$cfg -name '*kernel_trace.csv' -delete
  f=$(find $O/stats_$cfg -name '*kernel_stats.csv' | head -1)
  python3 - "$f" $cfg <<'PY' | tee -a $O/kernel_stats.txt
import csv,sys
print("==", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if 'lq::k_batch' in r['Name']: print("%-44s calls %s avg %.2f min %.2f max %.2f us" % (r['Name'][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
for lib in P1 P2; do
  LQ_HIP_LIB=$CS/liblq_hip_dev_$lib.so LQ_TIMELINE_STORAGE=oihw timeout -k 10 120 python3 tools/block_timeline.py imagenette:channelwise bwd > $O/timeline_bwd_$lib.txt 2>&1
done
head -20 $O/timeline_bwd_P1.txt
