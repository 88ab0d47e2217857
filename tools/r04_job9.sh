#!/bin/bash
# round 4, GPU job 9: batch fuzz over odd conv shapes; the pinned-head experiment; end-to-end steps; the bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job9
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_batch.py -q -m gpu > $O/pytest_batch.txt 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_batch.txt
CS=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc
BW="python3 tools/bench_weights.py --abi-only --kernel-storage oihw --steps 300"
for rep in 1 2; do
for lib in S PIN; do
  for cfg in imagenette:channelwise resnet50:channelwise imagenette:scalar; do
    LQ_HIP_LIB=$CS/liblq_hip_dev_$lib.so timeout -k 10 120 $BW --only $cfg 2>>$O/sweep.err | grep '^{' | sed "s/^{/{\"lib\": \"$lib\", /" >> $O/sweep.jsonl
  done
done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job9/sweep.jsonl"):
    r=json.loads(l)
    print(r["lib"], r["config"], r["orientation"], "abi %.1f  fused %.1f" % (r["us_per_step_batched_abi"], r["us_per_step_batched_abi_oihw_fused_update"]))
PY
LQ_HIP_LIB=$CS/liblq_hip_dev_S.so LQ_TIMELINE_STORAGE=oihw timeout -k 10 120 python3 tools/block_timeline.py imagenette:channelwise bwd > $O/timeline_bwd_S.txt 2>&1
LQ_HIP_LIB=$CS/liblq_hip_dev_PIN.so LQ_TIMELINE_STORAGE=oihw timeout -k 10 120 python3 tools/block_timeline.py imagenette:channelwise bwd > $O/timeline_bwd_PIN.txt 2>&1
grep -A20 "block range" $O/timeline_bwd_S.txt | head -24
bash tools/r04_evidence.sh $O e2e bench > $O/e2e_bench_log.txt 2>&1
tail -40 $O/e2e_bench_log.txt | cut -c1-250
