#!/bin/bash
# round 4, GPU job 22: relative placement of the streams of the BENCH launches
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_job22
mkdir -p $O
timeout -k 10 600 python3 tools/stream_offset_probe.py > $O/stream_offset_probe.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
cat $O/stream_offset_probe.txt
