#!/bin/bash
# round 4, GPU job 20: block timelines of k_batch_traverse_fin (from the tickets), with the task table
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job20
mkdir -p $O
CS=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc
for ab in 128; do
  LQ_TUNE_DUMP_TABLE=1 LQ_HIP_LIB=$CS/liblq_hip_dev.so LQ_TIMELINE_STORAGE=oihw LQ_TIMELINE_ABLATE=$ab timeout -k 10 200 python3 tools/block_timeline.py imagenette:channelwise bwd > $O/timeline_ab$ab.txt 2>$O/err_$ab.txt || exit 1
  head -30 $O/timeline_ab$ab.txt
  grep "lq dev" $O/err_$ab.txt
done
