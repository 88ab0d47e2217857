#!/bin/bash
# round 4, GPU job 2: (a) parity of the fragment column tiles (batch / layout / ddp tests, shipped library),
# (b) wall-clock sweep of the batch step over the compile-time variants (rows in flight per wave) x resident-block targets
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job2
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_layout.py tests/test_gpu_harness.py -q -m gpu -x > $O/pytest_subset.txt 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_subset.txt
CS=learned_quantization_amd/csrc
for lib in C A B; do
  for cfg in imagenette resnet50; do
    for nb in 1024 1280 1536 1792 2048 2560; do
      LQ_HIP_LIB=$GRAFT_REPO_ROOT/$CS/liblq_hip_dev_$lib.so LQ_TUNE_BATCH_NB=$nb timeout -k 10 120 python3 tools/bench_weights.py --only $cfg:channelwise --abi-only --kernel-storage oihw --steps 300 2>>$O/sweep.err | grep '^{' | sed "s/^{/{\"lib\": \"$lib\", \"nb\": $nb, /" >> $O/sweep.jsonl
    done
  done
  echo "lib $lib done" | tee -a $O/status.txt
done
# the per-column layout (no fragments) on the best-guess variant, and the old shipped numbers for reference come from profiles/r03
for cfg in imagenette resnet50; do
  LQ_HIP_LIB=$GRAFT_REPO_ROOT/$CS/liblq_hip_dev_C.so LQ_TUNE_BATCH_NB=1280 LQ_TUNE_BATCH_FRAG=0 timeout -k 10 120 python3 tools/bench_weights.py --only $cfg:channelwise --abi-only --kernel-storage oihw --steps 300 2>>$O/sweep.err | grep '^{' | sed "s/^{/{\"lib\": \"C_nofrag\", \"nb\": 1280, /" >> $O/sweep.jsonl
done
python3 - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r04_job2/sweep.jsonl")]
for r in rows:
    print(r["lib"], r["nb"], r["config"], "abi %.1f  fused %.1f" % (r["us_per_step_batched_abi"], r["us_per_step_batched_abi_oihw_fused_update"]))
PY
