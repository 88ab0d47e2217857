#!/bin/bash
# round 4, GPU job 7: weight-set timings for every orientation on the default storage, both storages, ABI only
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job7
mkdir -p $O
for st in oihw hwio; do
  timeout -k 10 500 python3 tools/bench_weights.py --abi-only --companion-only --kernel-storage $st --steps 200 2>>$O/err.log | grep '^{' >> $O/sweep_$st.jsonl
  echo "$st rc=$?" | tee -a $O/status.txt
done
python3 - <<'PY'
import json
for st in ("oihw","hwio"):
    for l in open(f"gpurun_out/r04_job7/sweep_{st}.jsonl"):
        r=json.loads(l); print(st, r['config'],r['orientation'],r['elements'], "abi %.1f fused %.1f" % (r['us_per_step_batched_abi'], r['us_per_step_batched_abi_oihw_fused_update']), "frac %.3f" % (16*r['elements']/r['us_per_step_batched_abi_oihw_fused_update']/1e-6/8e12))
PY
mkdir -p $O/stats_row
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_row -- python3 tools/bench_weights.py --only imagenette:rowwise --abi-only --kernel-storage oihw --steps 200 > $O/stats_row/run.log 2>&1
find $O/stats_row -name '*kernel_trace.csv' -delete
f=$(find $O/stats_row -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'lq::k_batch' in r['Name']: print("%-44s calls %s avg %.2f min %.2f max %.2f us" % (r['Name'][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
