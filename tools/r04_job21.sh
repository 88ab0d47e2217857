#!/bin/bash
# round 4, GPU job 21: K1 on the BENCH tensor with two float4 per thread (development knob LQ_TUNE_K1_U2) against the one-shot form, interleaved
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job21
mkdir -p $O
CS=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc
for rep in 1 2 3; do
for u2 in 0 1; do
  d=$O/u2_${u2}_$rep; mkdir -p $d
  LQ_HIP_LIB=$CS/liblq_hip_dev.so LQ_TUNE_K1_U2=$u2 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/shape_case.py 256 3 50176 --iters 80 --ops k1 --sets 4 > $d/run.log 2>&1 || exit 1
  find $d -name '*kernel_trace.csv' -delete
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  grep "lq::k_" $f | awk -F'","' -v m=$u2 -v r=$rep '{printf "u2=%s rep%s %s calls=%s avg=%.2f us\n", m, r, substr($1,2,60), $2, $4/1000}'
done
done
