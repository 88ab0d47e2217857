#!/bin/bash
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
for rep in 1 2 3; do
for mode in rs flat; do
  if [ $mode = flat ]; then export LQ_TUNE_S2=4096; else unset LQ_TUNE_S2; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${mode}_$rep -- python3 tools/shape_case.py 256 3 50176 --iters 60 --ops k1 --sets 4 > $out/${mode}_$rep.log 2>&1 || exit 1
  f=$(ls $out/${mode}_$rep/*/*kernel_stats.csv | head -1)
  grep "lq::k_" $f | awk -F, -v m=$mode -v r=$rep '{printf "%s rep%s %s calls=%s avg=%.2f us min=%.2f\n", m, r, substr($1,1,48), $(NF-6), $(NF-4)/1000, $(NF-2)/1000}'
done
done
