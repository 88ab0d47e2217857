#!/bin/bash
# round 4, GPU job 5: access-pattern microbenchmark of the column tiles; leaf-mode parity; host cost of the eager batched step
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job5
mkdir -p $O
timeout -k 10 120 tools/colbench 5 200 0 > $O/colbench_same_alignment.txt 2>&1
echo "colbench rc=$?" | tee -a $O/status.txt
timeout -k 10 120 tools/colbench 5 200 1 > $O/colbench_offset_second_stream.txt 2>&1
cat $O/colbench_same_alignment.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_layout.py tests/test_gpu_harness.py tests/test_gpu_ddp.py -q -m gpu > $O/pytest_subset.txt 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest_subset.txt
for cfg in mnist:rowwise cifar:channelwise imagenette:channelwise resnet50:channelwise; do
  timeout -k 10 300 python3 tools/bench_weights.py --only $cfg --kernel-storage oihw --steps 200 2>>$O/host.err | grep '^{' >> $O/host_cost.jsonl
done
cat $O/host_cost.jsonl
