#!/bin/bash
# usage: tools/r02_evidence.sh <outdir> [part]   (GPU box, repo root) -- the measurements behind profiles/r02*: bench lines,
# rocprofv3 summaries of the default bench command, PMC traffic, the shape sweep with counters, weight sweeps, the graphed
# data-parallel step on a one-rank RCCL group, statistics kernels, kernel breakdown of the end-to-end step.
out=$1; part=${2:-all}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ $part = all ] || [ $part = bench ]; then
  timeout -k 10 400 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_steps20.json 2>> $out/bench_default.err || exit 1
  timeout -k 10 200 python3 bench.py --steps 300 --warmup 30 --no-extras --no-cpu-baseline > $out/bench_steps300.json 2>> $out/bench_default.err || exit 1
  timeout -k 10 200 python3 bench.py --variant fused --no-extras --no-cpu-baseline > $out/bench_fused.json 2>> $out/bench_default.err || exit 1
  timeout -k 10 200 python3 bench.py --force-dist --no-extras --no-cpu-baseline > $out/bench_force_dist_rccl_1rank.json 2>> $out/bench_default.err || exit 1
  mkdir -p $out/prof_default_cmd
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_default_cmd -- python3 bench.py > $out/prof_default_cmd/bench.json 2> $out/prof_default_cmd/err.log || exit 1
  bash tools/prof_pmc.sh $out/pmc > $out/pmc.log 2>&1 || exit 1
  echo "bench part done"
fi
if [ $part = all ] || [ $part = shapes ]; then
  bash tools/prof_shapes.sh $out/shapes > $out/shapes.log 2>&1 || exit 1
  echo "shapes part done"
fi
if [ $part = all ] || [ $part = e2e ]; then
  timeout -k 10 600 python3 tools/bench_weights.py --steps 100 > $out/weight_sweeps.jsonl 2> $out/weight_sweeps.err || exit 1
  timeout -k 10 200 python3 tools/bench_stats.py > $out/bench_stats.txt 2>&1 || exit 1
  for args in "" "--batched" "--batched --graph" "--force-dist --batched" "--force-dist --batched --graph" "--force-dist --batched --graph --graph-collectives" "--force-dist --batched --graph --ddp-mode B"; do
    timeout -k 10 200 python3 -m learned_quantization_amd.train --config cifar --batch 256 --steps 60 --warmup 15 $args 2>> $out/e2e.err | grep '^{' >> $out/e2e_cifar.jsonl
  done
  timeout -k 10 200 python3 -m learned_quantization_amd.train --config cifar --mode nqcl --loss maxbin --value 1e-11 --rate 1e-7 --batch 256 --steps 60 --warmup 15 --batched --graph 2>> $out/e2e.err | grep '^{' >> $out/e2e_cifar.jsonl
  timeout -k 10 300 python3 -m learned_quantization_amd.train --config resnet50 --value 1e-11 --value-coarse 1e-10 --batch 32 --steps 12 --warmup 4 --batched --graph 2>> $out/e2e.err | grep '^{' >> $out/e2e_cifar.jsonl
  timeout -k 10 300 python3 -m learned_quantization_amd.train --config imagenette --batch 64 --steps 12 --warmup 4 --batched --graph 2>> $out/e2e.err | grep '^{' >> $out/e2e_cifar.jsonl
  # the per-GPU batch sizes BASELINE.json's configs name: ResNet-18-like bs 256, CIFAR nested quantization + loss term 1024 / 8 = 128,
  # ResNet-50-like 2048 / 8 = 256 (both data-parallel configs also on the one-rank RCCL group)
  timeout -k 10 400 python3 -m learned_quantization_amd.train --config imagenette --batch 256 --steps 8 --warmup 3 --batched --graph 2>> $out/e2e.err | grep '^{' >> $out/e2e_cifar.jsonl
  timeout -k 10 200 python3 -m learned_quantization_amd.train --config cifar --mode nqcl --loss maxbin --value 1e-11 --rate 1e-7 --batch 128 --steps 60 --warmup 15 --batched --graph --force-dist 2>> $out/e2e.err | grep '^{' >> $out/e2e_cifar.jsonl
  timeout -k 10 600 python3 -m learned_quantization_amd.train --config resnet50 --value 1e-11 --value-coarse 1e-10 --batch 256 --steps 6 --warmup 3 --batched --graph --force-dist 2>> $out/e2e.err | grep '^{' >> $out/e2e_cifar.jsonl
  mkdir -p $out/prof_e2e_cifar
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_e2e_cifar -- python3 tools/e2e_knobs.py cifar 256 0 1 > $out/prof_e2e_cifar/run.log 2> $out/prof_e2e_cifar/err.log || exit 1
  echo "e2e part done"
fi
echo "evidence done"
