#!/bin/bash
# round 4, GPU job 11: elements per block of the forward's column tiles (development knob), shipped forward body
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job11
mkdir -p $O
CS=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc
BW="python3 tools/bench_weights.py --abi-only --kernel-storage oihw --steps 300"
for rep in 1 2; do
for fw in 0 4096 6144 8192 12288 16384; do
  for cfg in imagenette:channelwise resnet50:channelwise; do
    LQ_HIP_LIB=$CS/liblq_hip_dev_S.so LQ_TUNE_BATCH_FWD_W=$fw timeout -k 10 120 $BW --only $cfg 2>>$O/sweep.err | grep '^{' | sed "s/^{/{\"lib\": \"S\", \"fwd_w\": $fw, /" >> $O/sweep.jsonl
  done
done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job11/sweep.jsonl"):
    r=json.loads(l)
    print(r["fwd_w"], r["config"], "abi %.1f  fused %.1f" % (r["us_per_step_batched_abi"], r["us_per_step_batched_abi_oihw_fused_update"]))
PY
