#!/bin/bash
# round 4, GPU job 8: periodic float4 stream for narrow column matrices and 4096-element units for long rows in the batch:
# parity, then the orientation sweep on the default storage and per-kernel durations
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job8
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_layout.py tests/test_gpu_harness.py tests/test_gpu_ddp.py -q -m gpu > $O/pytest_subset.txt 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_subset.txt
timeout -k 10 500 python3 tools/bench_weights.py --abi-only --companion-only --kernel-storage oihw --steps 200 2>>$O/err.log | grep '^{' >> $O/sweep_oihw.jsonl
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job8/sweep_oihw.jsonl"):
    r=json.loads(l); print("oihw", r['config'],r['orientation'],r['elements'], "abi %.1f fused %.1f" % (r['us_per_step_batched_abi'], r['us_per_step_batched_abi_oihw_fused_update']), "frac %.3f" % (16*r['elements']/r['us_per_step_batched_abi_oihw_fused_update']/1e-6/8e12))
PY
for o in rowwise columnwise scalar; do
  mkdir -p $O/stats_$o
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$o -- python3 tools/bench_weights.py --only imagenette:$o --abi-only --kernel-storage oihw --steps 200 > $O/stats_$o/run.log 2>&1
  find $O/stats_$o -name '*kernel_trace.csv' -delete
  f=$(find $O/stats_$o -name '*kernel_stats.csv' | head -1)
  python3 - "$f" $o <<'PY' | tee -a $O/kernel_stats.txt
import csv,sys
print("== imagenette", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if 'lq::k_batch' in r['Name']: print("%-44s calls %s avg %.2f min %.2f max %.2f us" % (r['Name'][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
