#!/bin/bash
# round 4, GPU job 3: parity of the fragment tiles (whole batch / layout files), per-kernel durations, block timelines, ablations
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job3
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_layout.py -q -m gpu > $O/pytest_subset.txt 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_subset.txt
CS=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc
BW="python3 tools/bench_weights.py --abi-only --kernel-storage oihw --steps 300"
# (1) per-kernel durations of the shipped library
for cfg in imagenette resnet50; do
  mkdir -p $O/stats_$cfg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$cfg -- python3 tools/bench_weights.py --only $cfg:channelwise --abi-only --kernel-storage oihw --steps 200 > $O/stats_$cfg/run.log 2>&1
  f=$(find $O/stats_$cfg -name '*kernel_stats.csv' | head -1)
  echo "== $cfg" >> $O/kernel_stats.txt; grep -E "lq::|Name" "$f" | cut -d, -f1-8 | head -12 >> $O/kernel_stats.txt
  find $O/stats_$cfg -name '*kernel_trace.csv' -delete
done
cat $O/kernel_stats.txt
# (2) ablations (development library C = the shipped geometry): 1 math-free, 8 no epilogue, 9 both
for ab in 0 1 8 9; do
  for cfg in imagenette resnet50; do
    LQ_HIP_LIB=$CS/liblq_hip_dev_C.so timeout -k 10 120 $BW --only $cfg:channelwise --ablate $ab 2>>$O/abl.err | grep '^{' | sed "s/^{/{\"lib\": \"C\", \"ablate\": $ab, /" >> $O/ablate.jsonl
  done
done
# (3) rows in flight per wave = 8 (one round for 32-row blocks), several block sizes
for nb in 1024 1280 2048; do
  for cfg in imagenette resnet50; do
    LQ_HIP_LIB=$CS/liblq_hip_dev_D.so LQ_TUNE_BATCH_NB=$nb timeout -k 10 120 $BW --only $cfg:channelwise 2>>$O/abl.err | grep '^{' | sed "s/^{/{\"lib\": \"D\", \"nb\": $nb, /" >> $O/ablate.jsonl
  done
done
for w in 4096 16384 32768; do
  for lib in C D; do
    LQ_HIP_LIB=$CS/liblq_hip_dev_$lib.so LQ_TUNE_BATCH_W=$w timeout -k 10 120 $BW --only imagenette:channelwise 2>>$O/abl.err | grep '^{' | sed "s/^{/{\"lib\": \"$lib\", \"w\": $w, /" >> $O/ablate.jsonl
  done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job3/ablate.jsonl"):
    r=json.loads(l)
    print({k:r[k] for k in ("lib","ablate","nb","w") if k in r}, r["config"], "abi %.1f  fused %.1f" % (r["us_per_step_batched_abi"], r["us_per_step_batched_abi_oihw_fused_update"]))
PY
# (4) block timelines of the scale-gradient traversal
for lib in C A D; do
  LQ_HIP_LIB=$CS/liblq_hip_dev_$lib.so LQ_TIMELINE_STORAGE=oihw timeout -k 10 120 python3 tools/block_timeline.py imagenette:channelwise bwd > $O/timeline_bwd_$lib.txt 2>&1
done
LQ_HIP_LIB=$CS/liblq_hip_dev_C.so LQ_TIMELINE_STORAGE=oihw timeout -k 10 120 python3 tools/block_timeline.py imagenette:channelwise fwd > $O/timeline_fwd_C.txt 2>&1
head -30 $O/timeline_bwd_C.txt
