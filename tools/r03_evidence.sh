#!/bin/bash
# usage: tools/r03_evidence.sh <outdir> <step>...   (GPU box, repo root) -- round-3 evidence, one step per word
# (base_r02*: first `mkdir _base_r02 && git archive ad7e194 | tar -x -C _base_r02 && make -C _base_r02/learned_quantization_amd/csrc`):
#   tests      the GPU parity suite (plain)
#   batch      rocprofv3 stats + FETCH/WRITE/SQ counters of the multi-tensor batch on the ResNet-18-like / ResNet-50-like sets
#   base_r02   the same kernel stats + FETCH/WRITE of the round-2 sources (tree at ad7e194 extracted to _base_r02/, built there)
#   sweeps     tools/bench_weights.py over every config and orientation -> r03_weight_sweeps.jsonl
#   cov        kernel trace of the GPU parity suite -> kernel_coverage.txt
#   bench      bench.py default line
out=$1; shift
mkdir -p $out
for step in "$@"; do
case $step in
tests)
  timeout -k 10 900 python3 -m pytest tests -q -m gpu -x -p no:cacheprovider > $out/pytest_gpu.log 2>&1; rc=$?
  tail -n 4 $out/pytest_gpu.log; [ $rc -eq 0 ] || exit 1 ;;
batch)
  bash tools/prof_batch.sh $out/batch_imagenette_channelwise imagenette:channelwise || exit 1
  bash tools/prof_batch.sh $out/batch_resnet50_channelwise resnet50:channelwise || exit 1 ;;
batch_oihw_storage)
  # the same two weight sets with the conv kernels STORED in OIHW order (layers.py kernel_storage, the trainer's default since
  # round 3): plain streaming launches, no companion, dP = the weight gradient; finalize + Adam in one launch
  bash tools/prof_batch.sh $out/batch_oihw_storage_imagenette_channelwise imagenette:channelwise --kernel-storage oihw --companion-only || exit 1
  bash tools/prof_batch.sh $out/batch_oihw_storage_resnet50_channelwise resnet50:channelwise --kernel-storage oihw --companion-only || exit 1 ;;
base_r02)
  ( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT/_base_r02
    for only in imagenette:channelwise resnet50:channelwise; do
      d=$GRAFT_REPO_ROOT/$out/base_r02_$(echo $only | tr ':' '_'); mkdir -p $d/stats
      timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d/stats -- python3 tools/bench_weights.py --only $only > $d/stats/run.log 2> $d/stats/err.log || exit 1
      for c in FETCH_SIZE WRITE_SIZE; do
        mkdir -p $d/$c
        timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d/$c -- python3 tools/bench_weights.py --only $only --steps 20 > $d/$c/run.log 2> $d/$c/err.log || exit 1
      done
      python3 $GRAFT_REPO_ROOT/tools/prof_batch_summary.py $d $only > $d/summary.txt; cat $d/summary.txt
    done ) || exit 1 ;;
base_r02_oihw)
  ( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT/_base_r02
    for only in imagenette:channelwise resnet50:channelwise; do
      d=$GRAFT_REPO_ROOT/$out/base_r02_oihw_$(echo $only | tr ':' '_'); mkdir -p $d
      timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 ../tools/base_r02_oihw.py $only > $d/run.log 2> $d/err.log || { tail -n 5 $d/err.log; exit 1; }
      cat $d/run.log
      python3 - "$(ls -S $d/*/*_kernel_stats.csv | head -n 1)" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_batch_traverse" in r["Name"]:
        print(f'  {r["Name"].split("(")[0].replace("void ","")[:40]:40s} calls={r["Calls"]:>5s} avg={float(r["AverageNs"])/1e3:7.2f} min={float(r["MinNs"])/1e3:7.2f}')
PY
    done ) || exit 1 ;;
sweeps)
  timeout -k 10 600 python3 tools/bench_weights.py > $out/r03_weight_sweeps.jsonl 2> $out/sweeps.err || exit 1
  cat $out/r03_weight_sweeps.jsonl ;;
cov)
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  mkdir -p $out/cov
  timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cov -- python3 -m pytest tests -q -m gpu -x --deselect tests/test_gpu_ddp.py -p no:cacheprovider > $out/cov/pytest.log 2> $out/cov/err.log
  echo "pytest rc=$?"; tail -n 3 $out/cov/pytest.log
  python3 tools/kernel_coverage.py $out/cov > $out/kernel_coverage.txt; echo "coverage rc=$?"
  find $out/cov -name "*kernel_trace.csv" -size +4M -delete
  head -n 60 $out/kernel_coverage.txt ;;
exchange)
  # the ds exchange of the bench step on a one-rank RCCL communicator: graph without / with the captured all-reduce, eager forms
  for v in "--graph" "--graph --force-dist" "--force-dist --exchange sync" "--force-dist --exchange async" ""; do
    n=$(echo "bench$v" | tr -d ' ' | tr '-' '_')
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras $v > $out/$n.json 2> $out/$n.err || { tail -n 5 $out/$n.err; exit 1; }
    python3 -c "import json,sys; l=json.load(open('$out/$n.json')); print('$n', round(l['ms_per_step']*1e3,2), 'us/step', l['config'].get('launch'), '|', l['config'].get('exchange'))"
  done ;;
exchange2)
  # which cross-branch edges of the captured exchange cost what (one-rank RCCL communicator)
  for v in "--graph" "--graph --force-dist --graph-edges fork_join" "--graph --force-dist --graph-edges fork_only" "--graph --force-dist --graph-edges linear" "--force-dist --exchange sync"; do
    n=$(echo "bench$v" | tr -d ' ' | tr '-' '_')
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras $v > $out/$n.json 2> $out/$n.err || { tail -n 5 $out/$n.err; exit 1; }
    python3 -c "import json,sys; l=json.loads([x for x in open('$out/$n.json') if x.startswith('{')][-1]); print('$n', round(l['ms_per_step']*1e3,2), 'us/step', l['config'].get('launch'), '|', l['config'].get('exchange'))"
  done
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  for e in fork_join linear; do
    mkdir -p $out/trace_$e
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$e -- python3 bench.py --no-cpu-baseline --no-extras --graph --force-dist --graph-edges $e --steps 64 --warmup 16 > $out/trace_$e/run.log 2> $out/trace_$e/err.log || exit 1
    python3 - $out/trace_$e <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:8]:
    print("  ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
PY
  done ;;
timeline)
  export LQ_HIP_LIB=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc/liblq_hip_dev.so
  for c in imagenette:channelwise resnet50:channelwise; do
    for w in fwd bwd_oihw; do
      n=$(echo "timeline_${c}_$w" | tr ':' '_')
      timeout -k 10 200 python3 tools/block_timeline.py $c $w > $out/$n.txt 2> $out/$n.err || { tail -n 5 $out/$n.err; exit 1; }
      head -n 2 $out/$n.txt
    done
  done
  LQ_TIMELINE_HWIO_OUT=0 timeout -k 10 200 python3 tools/block_timeline.py imagenette:channelwise fwd > $out/timeline_imagenette_channelwise_fwd_companion_only.txt 2>> $out/timeline.err || exit 1
  # conv kernels stored OIHW (the trainer's default): plain streaming launches
  for c in imagenette:channelwise resnet50:channelwise; do
    for w in fwd bwd; do
      n=$(echo "timeline_oihw_storage_${c}_$w" | tr ':' '_')
      LQ_TIMELINE_STORAGE=oihw LQ_TIMELINE_HWIO_OUT=0 timeout -k 10 200 python3 tools/block_timeline.py $c $w > $out/$n.txt 2> $out/$n.err || { tail -n 5 $out/$n.err; exit 1; }
      head -n 2 $out/$n.txt
    done
  done
  unset LQ_HIP_LIB ;;
batchtests)
  timeout -k 10 900 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_harness.py -q -m gpu -x -p no:cacheprovider > $out/pytest_batch.log 2>&1; rc=$?
  tail -n 15 $out/pytest_batch.log; [ $rc -eq 0 ] || exit 1 ;;
quick)
  # wall-clock + rocprofv3 kernel stats of the batch ABI on the two ResNet sets, both forward forms
  bash tools/prof_batch_quick.sh $out/quick imagenette:channelwise resnet50:channelwise || exit 1
  bash tools/prof_batch_quick.sh $out/quick_companion_only imagenette:channelwise resnet50:channelwise -- --companion-only || exit 1 ;;
newtests)
  timeout -k 10 900 python3 -m pytest tests/test_gpu_ddp.py tests/test_gpu_layers.py -q -m gpu -x -p no:cacheprovider > $out/pytest_new.log 2>&1; rc=$?
  tail -n 15 $out/pytest_new.log; [ $rc -eq 0 ] || exit 1 ;;
pmc)
  # HBM traffic of K1 / K2 / K4 on the BENCH tensor (FETCH_SIZE / WRITE_SIZE passes) -> traffic.json tied to the kernel sources
  bash tools/prof_pmc.sh $out/pmc || exit 1
  cat $out/pmc/traffic.json | head -n 5 ;;
stats)
  # rocprofv3 --kernel-trace --stats of exactly the default bench command
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  mkdir -p $out/default_cmd
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/default_cmd -- python3 bench.py > $out/default_cmd/bench.json 2> $out/default_cmd/err.log || exit 1
  python3 - "$(find $out/default_cmd -name '*kernel_stats.csv' | head -n 1)" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lq::" in r["Name"]:
        print(f'{r["Name"][:74]:74s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.2f}')
PY
  ;;
membench)
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/membench tools/membench.hip || exit 1
  timeout -k 10 300 /tmp/membench > $out/membench.txt 2>&1 || exit 1
  grep "BS512" $out/membench.txt | head -n 12 ;;
e2e)
  # end-to-end steps of the BASELINE configs (synthetic data), as profiles/r02_e2e.jsonl: eager / batched / graphed, the one-rank
  # RCCL group (the one-graph step with the captured all-reduce is the default graphed data-parallel form since round 3), mode B,
  # nested quantization + loss term, the ResNets at the per-GPU batch sizes of BASELINE.json
  rm -f $out/r03_e2e.jsonl
  run() { timeout -k 10 600 python3 -m learned_quantization_amd.train "$@" 2>> $out/e2e.err | grep '^{' >> $out/r03_e2e.jsonl || { tail -n 5 $out/e2e.err; exit 1; }; }
  for args in "" "--batched" "--batched --graph" "--force-dist --batched" "--force-dist --batched --graph --no-graph-collectives" "--force-dist --batched --graph" "--force-dist --batched --graph --ddp-mode B"; do
    run --config cifar --batch 256 --steps 60 --warmup 15 $args
  done
  run --config cifar --mode nqcl --loss maxbin --value 1e-11 --rate 1e-7 --batch 256 --steps 60 --warmup 15 --batched --graph
  run --config cifar --mode nqcl --loss maxbin --value 1e-11 --rate 1e-7 --batch 128 --steps 60 --warmup 15 --batched --graph --force-dist
  run --config cifar --mode nqcl --loss maxbin --value 1e-11 --rate 1e-7 --batch 128 --steps 60 --warmup 15 --batched --graph --force-dist --ddp-mode B
  run --config imagenette --batch 64 --steps 12 --warmup 4 --batched --graph
  run --config imagenette --batch 64 --steps 12 --warmup 4 --batched --graph --kernel-storage hwio
  run --config imagenette --batch 256 --steps 8 --warmup 3 --batched --graph
  run --config imagenette --batch 256 --steps 8 --warmup 3 --batched --graph --force-dist --ddp-mode B
  run --config resnet50 --value 1e-11 --value-coarse 1e-10 --batch 32 --steps 12 --warmup 4 --batched --graph
  run --config resnet50 --value 1e-11 --value-coarse 1e-10 --batch 256 --steps 6 --warmup 3 --batched --graph --force-dist
  python3 - $out/r03_e2e.jsonl <<'PY'
import sys,json
for l in open(sys.argv[1]):
    d=json.loads(l); print(d.get('config'), d.get('mode'), d.get('per_gpu_batch'), 'graph' if d.get('hipgraph') else 'eager', 'one-graph' if d.get('graph_collectives') else '', 'batched' if d.get('batched') else '', d.get('backend'), d.get('ddp_mode'), round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms')
PY
  ;;
sweeps_companion)
  timeout -k 10 300 python3 tools/bench_weights.py --abi-only --companion-only > $out/r03_weight_sweeps_companion_only.jsonl 2>> $out/sweeps.err || exit 1
  cat $out/r03_weight_sweeps_companion_only.jsonl ;;
bench)
  timeout -k 10 400 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
  cat $out/bench_default.json ;;
esac
done
