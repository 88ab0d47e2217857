#!/bin/bash
# round 4, GPU job 15: the periodic float4 stream with the scales requested first and the contexts formed behind the first loads:
# parity (single-tensor and batch files), then OLD against NEW on the row-wise / column-wise weight sets
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job15
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_instantiations.py tests/test_gpu_fuzz.py tests/test_gpu_batch.py tests/test_gpu_layout.py -q -m gpu -x > $O/pytest_subset.txt 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_subset.txt
CS=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc
BW="python3 tools/bench_weights.py --abi-only --kernel-storage oihw --steps 300"
for rep in 1 2; do
for lib in OLD NEW; do
  for cfg in imagenette:rowwise imagenette:columnwise resnet50:rowwise cifar:rowwise; do
    LQ_HIP_LIB=$CS/liblq_hip_dev_$lib.so timeout -k 10 120 $BW --only $cfg 2>>$O/sweep.err | grep '^{' | sed "s/^{/{\"lib\": \"$lib\", /" >> $O/sweep.jsonl
  done
done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job15/sweep.jsonl"):
    r=json.loads(l)
    print(r["lib"], r["config"], r["orientation"], "abi %.1f  fused %.1f" % (r["us_per_step_batched_abi"], r["us_per_step_batched_abi_oihw_fused_update"]))
PY
