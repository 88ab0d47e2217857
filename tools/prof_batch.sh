#!/bin/bash
# usage: tools/prof_batch.sh <outdir> <config:orientation> [bench_weights args]  (GPU box, repo root)
# rocprofv3 evidence for the multi-tensor batch kernels on one weight set: kernel-trace stats, then FETCH_SIZE / WRITE_SIZE
# and the SQ split in separate --pmc passes (kernel-trace only beside them); tools/bench_weights.py stands directly behind `--`.
out=$1; only=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out/stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 tools/bench_weights.py --only $only --abi-only "$@" > $out/stats/run.log 2> $out/stats/err.log || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $out/$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 tools/bench_weights.py --only $only --abi-only --steps 20 "$@" > $out/$c/run.log 2> $out/$c/err.log || exit 1
done
mkdir -p $out/SQ
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $out/SQ -- python3 tools/bench_weights.py --only $only --abi-only --steps 20 "$@" > $out/SQ/run.log 2> $out/SQ/err.log || exit 1
python3 tools/prof_batch_summary.py $out $only > $out/summary.txt
find $out -name "*kernel_trace.csv" -size +1M -delete      # the summaries are what is kept (gpurun merges at most 64 MiB back)
cat $out/summary.txt
