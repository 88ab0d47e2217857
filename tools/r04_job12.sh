#!/bin/bash
# round 4, GPU job 12: leaf-mode host fast paths: parity (batch / layout / harness / ddp files) and the host cost of the eager forms
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job12
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_layout.py tests/test_gpu_harness.py tests/test_gpu_ddp.py -q -m gpu -x > $O/pytest_subset.txt 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_subset.txt
for cfg in mnist:rowwise cifar:channelwise imagenette:channelwise resnet50:channelwise; do
  timeout -k 10 300 python3 tools/bench_weights.py --only $cfg --kernel-storage oihw --steps 200 2>>$O/host.err | grep '^{' >> $O/host_cost.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job12/host_cost.jsonl"):
    r=json.loads(l); print(r['config'],{k[12:]:round(v,1) for k,v in r.items() if k.startswith('us_')})
PY
for args in "--batched" "--force-dist --batched" "--force-dist --batched --ddp-mode B"; do
  timeout -k 10 300 python3 -m learned_quantization_amd.train --config cifar --batch 256 --steps 60 --warmup 15 $args 2>>$O/e2e.err | grep '^{' >> $O/e2e.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job12/e2e.jsonl"):
    d=json.loads(l); print(d.get('config'), 'eager batched', d.get('backend'), d.get('ddp_mode'), round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms')
PY
