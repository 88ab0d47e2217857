#!/bin/bash
# usage: tools/r04_evidence.sh <outdir> <step>...   (GPU box, repo root) -- round-4 evidence, one step per word:
#   tests   the GPU suite, plain (-x, the driver's command)
#   cov     kernel trace of the GPU suite -> kernel_coverage.txt (every shipped instantiation launched by a parity case)
#   pmc     FETCH_SIZE / WRITE_SIZE of the BENCH kernels -> pmc/traffic.json (bench.py reads profiles/traffic.json)
#   stats   rocprofv3 --kernel-trace --stats of the default bench command
#   batch   tools/prof_batch.sh on the ResNet-18-like / ResNet-50-like weight sets, kernels stored OIHW: stats, FETCH/WRITE, SQ
#   sweeps  tools/bench_weights.py, every config and orientation, default storage (ABI only) and host cost of the eager forms
#   e2e     end-to-end steps of the BASELINE configs (synthetic data)
#   bench   bench.py default line
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for step in "$@"; do
case $step in
tests)
  timeout -k 10 900 python3 -m pytest tests -q -m gpu -x -p no:cacheprovider > $out/pytest_gpu.log 2>&1; rc=$?
  tail -n 4 $out/pytest_gpu.log; [ $rc -eq 0 ] || exit 1 ;;
cov)
  mkdir -p $out/cov
  timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cov -- python3 -m pytest tests -q -m gpu -x --deselect tests/test_gpu_ddp.py -p no:cacheprovider > $out/cov/pytest.log 2> $out/cov/err.log
  echo "pytest rc=$?"; tail -n 3 $out/cov/pytest.log
  python3 tools/kernel_coverage.py $out/cov > $out/kernel_coverage.txt; echo "coverage rc=$?"
  find $out/cov -name "*kernel_trace.csv" -size +4M -delete
  head -n 40 $out/kernel_coverage.txt ;;
pmc)
  bash tools/prof_pmc.sh $out/pmc || exit 1
  find $out/pmc -name "*.csv" -size +4M -delete ;;
stats)
  bash tools/prof_default_cmd.sh > $out/default_cmd_stats.txt 2>&1
  cp gpurun_out/prof_default_cmd/bench.json $out/bench_line_under_rocprofv3_default_cmd.json 2>/dev/null
  f=$(find gpurun_out/prof_default_cmd -name "*kernel_stats.csv" | head -1); cp "$f" $out/rocprofv3_kernel_stats_default_cmd.csv
  find gpurun_out/prof_default_cmd -name "*kernel_trace.csv" -delete
  cat $out/default_cmd_stats.txt | head -30 ;;
batch)
  bash tools/prof_batch.sh $out/batch_oihw_storage_imagenette_channelwise imagenette:channelwise --kernel-storage oihw --companion-only || exit 1
  bash tools/prof_batch.sh $out/batch_oihw_storage_resnet50_channelwise resnet50:channelwise --kernel-storage oihw --companion-only || exit 1
  bash tools/prof_batch.sh $out/batch_oihw_storage_imagenette_rowwise imagenette:rowwise --kernel-storage oihw --companion-only || exit 1
  bash tools/prof_batch.sh $out/batch_oihw_storage_imagenette_scalar imagenette:scalar --kernel-storage oihw --companion-only || exit 1 ;;
sweeps)
  timeout -k 10 600 python3 tools/bench_weights.py --abi-only --companion-only --kernel-storage oihw > $out/r04_weight_sweeps_oihw_storage.jsonl 2> $out/sweeps.err || exit 1
  timeout -k 10 600 python3 tools/bench_weights.py --abi-only --companion-only --kernel-storage hwio > $out/r04_weight_sweeps_hwio_storage.jsonl 2>> $out/sweeps.err || exit 1
  for cfg in mnist:rowwise cifar:channelwise imagenette:channelwise resnet50:channelwise; do
    timeout -k 10 300 python3 tools/bench_weights.py --only $cfg --kernel-storage oihw --steps 200 2>> $out/sweeps.err | grep '^{' >> $out/r04_host_cost_eager_forms.jsonl
  done
  python3 - $out <<'PY'
import json,sys
for st in ("oihw","hwio"):
    for l in open(f"{sys.argv[1]}/r04_weight_sweeps_{st}_storage.jsonl"):
        r=json.loads(l); t=r['us_per_step_batched_abi_oihw_fused_update']
        print(st, r['config'],r['orientation'],r['elements'], "abi %.1f fused %.1f us" % (r['us_per_step_batched_abi'], t), "frac of 8 TB/s %.3f" % (16*r['elements']/t/1e-6/8e12))
for l in open(f"{sys.argv[1]}/r04_host_cost_eager_forms.jsonl"):
    r=json.loads(l); print(r['config'],r['orientation'],{k:round(v,1) for k,v in r.items() if k.startswith('us_')})
PY
  ;;
e2e)
  rm -f $out/r04_e2e.jsonl
  run() { timeout -k 10 600 python3 -m learned_quantization_amd.train "$@" 2>> $out/e2e.err | grep '^{' >> $out/r04_e2e.jsonl || { tail -n 5 $out/e2e.err; exit 1; }; }
  for args in "" "--batched" "--batched --graph" "--force-dist --batched" "--force-dist --batched --ddp-mode B" "--force-dist --batched --graph --no-graph-collectives" "--force-dist --batched --graph" "--force-dist --batched --graph --ddp-mode B"; do
    run --config cifar --batch 256 --steps 60 --warmup 15 $args
  done
  run --config mnist --orientation rowwise --value 1e-10 --batch 128 --steps 100 --warmup 20 --batched
  run --config mnist --orientation rowwise --value 1e-10 --batch 128 --steps 100 --warmup 20 --batched --graph
  run --config cifar --mode nqcl --loss maxbin --value 1e-11 --rate 1e-7 --batch 256 --steps 60 --warmup 15 --batched --graph
  run --config cifar --mode nqcl --loss maxbin --value 1e-11 --rate 1e-7 --batch 128 --steps 60 --warmup 15 --batched --graph --force-dist
  run --config imagenette --batch 64 --steps 12 --warmup 4 --batched --graph
  run --config imagenette --batch 256 --steps 8 --warmup 3 --batched --graph
  run --config imagenette --batch 256 --steps 8 --warmup 3 --batched --graph --force-dist --ddp-mode B
  run --config resnet50 --value 1e-11 --value-coarse 1e-10 --batch 32 --steps 12 --warmup 4 --batched --graph
  python3 - $out/r04_e2e.jsonl <<'PY'
import sys,json
for l in open(sys.argv[1]):
    d=json.loads(l); print(d.get('config'), d.get('mode'), d.get('per_gpu_batch'), 'graph' if d.get('hipgraph') else 'eager', 'one-graph' if d.get('graph_collectives') else '', 'batched' if d.get('batched') else '', d.get('backend'), d.get('ddp_mode'), round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms')
PY
  ;;
bench)
  timeout -k 10 400 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
  tail -c 1500 $out/bench_default.json ;;
esac
done
