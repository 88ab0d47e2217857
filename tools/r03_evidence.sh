#!/bin/bash
# usage: tools/r03_evidence.sh <outdir> <step>...   (GPU box, repo root) -- round-3 evidence, one step per word:
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
bench)
  timeout -k 10 400 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
  cat $out/bench_default.json ;;
esac
done
