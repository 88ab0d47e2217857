#!/bin/bash
# round 4, GPU job 16: KerasAdam.step host fast path: the suite files that use it, eager end-to-end steps
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job16
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_layers.py tests/test_gpu_layout.py tests/test_gpu_batch.py tests/test_gpu_harness.py tests/test_gpu_ddp.py -q -m gpu -x > $O/pytest_subset.txt 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_subset.txt
for args in "--config cifar --batch 256 --steps 80 --warmup 20 --batched" "--config cifar --batch 256 --steps 80 --warmup 20 --batched --graph" "--config cifar --batch 256 --steps 80 --warmup 20 --batched --force-dist" "--config mnist --orientation rowwise --value 1e-10 --batch 128 --steps 200 --warmup 40 --batched" "--config imagenette --batch 64 --steps 12 --warmup 4 --batched" "--config imagenette --batch 64 --steps 12 --warmup 4 --batched --graph"; do
  timeout -k 10 300 python3 -m learned_quantization_amd.train $args 2>>$O/e2e.err | grep '^{' >> $O/e2e.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job16/e2e.jsonl"):
    d=json.loads(l); print(d.get('config'), d.get('per_gpu_batch'), 'graph' if d.get('hipgraph') else 'eager', d.get('backend'), round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms')
PY
