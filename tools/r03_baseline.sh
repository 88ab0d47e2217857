#!/bin/bash
# usage: tools/r03_baseline.sh <outdir>  (GPU box, repo root) -- round-3 evidence taken BEFORE the batch kernels were reworked:
# rocprofv3 stats + counters of the multi-tensor batch on the ResNet-18-like / ResNet-50-like weight sets, and a kernel trace
# of the GPU parity suite for tools/kernel_coverage.py.
out=$1
mkdir -p $out
bash tools/prof_batch.sh $out/batch_imagenette_channelwise imagenette:channelwise || exit 1
bash tools/prof_batch.sh $out/batch_resnet50_channelwise resnet50:channelwise || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out/cov
timeout -k 10 800 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cov -- python3 -m pytest tests -q -m gpu -x --deselect tests/test_gpu_ddp.py -p no:cacheprovider > $out/cov/pytest.log 2> $out/cov/err.log
echo "pytest rc=$?"
tail -3 $out/cov/pytest.log
python3 tools/kernel_coverage.py $out/cov > $out/kernel_coverage.txt; echo "coverage rc=$?"
head -40 $out/kernel_coverage.txt
# keep only the stats csv of the coverage run (the per-dispatch trace is large)
find $out/cov -name "*kernel_trace.csv" -size +4M -delete
