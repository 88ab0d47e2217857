#!/bin/bash
# round 4, GPU job 6: self-contained finalize records; per-column form / nontemporal loads; leaf-mode host cost after the fast path
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job6
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_layout.py tests/test_gpu_harness.py tests/test_gpu_ddp.py -q -m gpu > $O/pytest_subset.txt 2>&1
echo "pytest rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_subset.txt
CS=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc
BW="python3 tools/bench_weights.py --abi-only --kernel-storage oihw --steps 300"
for lib in S G4 NT G4NT; do
  for cfg in imagenette resnet50; do
    for nb in 1024 1280; do
      LQ_HIP_LIB=$CS/liblq_hip_dev_$lib.so LQ_TUNE_BATCH_NB=$nb timeout -k 10 120 $BW --only $cfg:channelwise 2>>$O/sweep.err | grep '^{' | sed "s/^{/{\"lib\": \"$lib\", \"nb\": $nb, /" >> $O/sweep.jsonl
    done
  done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job6/sweep.jsonl"):
    r=json.loads(l)
    print({k:r[k] for k in ("lib","ablate","nb","w") if k in r}, r["config"], "abi %.1f  fused %.1f" % (r["us_per_step_batched_abi"], r["us_per_step_batched_abi_oihw_fused_update"]))
PY
for cfg in imagenette resnet50; do
  mkdir -p $O/stats_$cfg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$cfg -- python3 tools/bench_weights.py --only $cfg:channelwise --abi-only --kernel-storage oihw --steps 200 > $O/stats_$cfg/run.log 2>&1
  find $O/stats_$cfg -name '*kernel_trace.csv' -delete
  f=$(find $O/stats_$cfg -name '*kernel_stats.csv' | head -1)
  python3 - "$f" $cfg <<'PY' | tee -a $O/kernel_stats.txt
import csv,sys
print("==", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if 'lq::k_batch' in r['Name']: print("%-44s calls %s avg %.2f min %.2f max %.2f us" % (r['Name'][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
for cfg in cifar:channelwise imagenette:channelwise resnet50:channelwise; do
  timeout -k 10 300 python3 tools/bench_weights.py --only $cfg --kernel-storage oihw --steps 200 2>>$O/host.err | grep '^{' >> $O/host_cost.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_job6/host_cost.jsonl"):
    r=json.loads(l); print(r['config'],r['orientation'],{k:round(v,1) for k,v in r.items() if k.startswith('us_')})
PY
