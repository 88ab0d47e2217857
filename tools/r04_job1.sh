#!/bin/bash
# round 4, GPU job 1: (a) what differed in GPUTEST_r03's red test (kernel trace of the three ResNet-18 runs), (b) the full GPU suite
# in the new order, without -x, so that every failure is seen at once
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job1
mkdir -p $O
timeout -k 10 420 rocprofv3 --kernel-trace --output-format csv -d $O/diag_trace -- python3 tools/r04_rehearsal_diag.py 29551 > $O/diag.log 2>&1
echo "diag rc=$?" | tee -a $O/status.txt
python3 tools/r04_rehearsal_diag_kernels.py $O/diag_trace > $O/diag_kernels.txt 2>&1
echo "diag_kernels rc=$?" | tee -a $O/status.txt
# the traces are large: keep the summary only
find $O/diag_trace -name '*.csv' -size +2M -delete
timeout -k 10 900 python3 -m pytest tests -q -m gpu --durations=15 > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -5 $O/pytest_gpu.txt
