#!/bin/bash
# round 4, GPU job 10: the forward's column tiles with the addressing of lq_batch_cols.hpp (compile-time experiment FT) against the shipped S
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job10
mkdir -p $O
CS=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc
BW="python3 tools/bench_weights.py --abi-only --kernel-storage oihw --steps 300"
for rep in 1 2 3; do
for lib in S FT; do
  for cfg in imagenette:channelwise resnet50:channelwise; do
    LQ_HIP_LIB=$CS/liblq_hip_dev_$lib.so timeout -k 10 120 $BW --only $cfg 2>>$O/sweep.err | grep '^{' | sed "s/^{/{\"lib\": \"$lib\", /" >> $O/sweep.jsonl
  done
done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job10/sweep.jsonl"):
    r=json.loads(l)
    print(r["lib"], r["config"], r["orientation"], "abi %.1f  fused %.1f" % (r["us_per_step_batched_abi"], r["us_per_step_batched_abi_oihw_fused_update"]))
PY
for lib in S FT; do
  mkdir -p $O/stats_$lib
  LQ_HIP_LIB=$CS/liblq_hip_dev_$lib.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$lib -- python3 tools/bench_weights.py --only imagenette:channelwise --abi-only --kernel-storage oihw --steps 200 > $O/stats_$lib/run.log 2>&1
  find $O/stats_$lib -name '*kernel_trace.csv' -delete
  f=$(find $O/stats_$lib -name '*kernel_stats.csv' | head -1)
  python3 - "$f" $lib <<'PY'
import csv,sys
print("==", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if 'lq::k_batch' in r['Name']: print("%-44s calls %s avg %.2f min %.2f max %.2f us" % (r['Name'][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
# parity of the experiment build: the batch file through the FT library
LQ_HIP_LIB=$CS/liblq_hip_dev_FT.so timeout -k 10 300 python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from learned_quantization_amd import _hip
_hip.use_library(os.environ["LQ_HIP_LIB"])
import pytest
sys.exit(pytest.main(["tests/test_gpu_batch.py", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"]))
PY
echo "FT parity rc=$?"
