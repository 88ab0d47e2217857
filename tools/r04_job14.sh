#!/bin/bash
# round 4, GPU job 14: the two-rank hold test, and its mutation check (with the hold removed it must fail)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job14
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ddp.py -q -m gpu -k "regularised_kernels_are_exchanged" > $O/hold_test.txt 2>&1
echo "hold test rc=$? (expected 0)" | tee -a $O/status.txt
LQ_TEST_MUTATE_NO_HOLD=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_ddp.py -q -m gpu -k "regularised_kernels_are_exchanged" > $O/hold_test_mutated.txt 2>&1
echo "mutated rc=$? (expected 1: the test must detect a missing hold)" | tee -a $O/status.txt
grep -E "assert|Error|passed|failed" $O/hold_test_mutated.txt | tail -5
timeout -k 10 900 python3 -m pytest tests/test_gpu_ddp.py -q -m gpu > $O/ddp_file.txt 2>&1
echo "ddp file rc=$?" | tee -a $O/status.txt
tail -3 $O/ddp_file.txt
