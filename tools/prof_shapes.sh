#!/bin/bash
# usage: tools/prof_shapes.sh <outdir> [case ...]      (run on the GPU box, from the repo root)
# rocprofv3 evidence for the non-BENCH traversal modes: per case one `--kernel-trace --stats` pass and three `--pmc`
# passes (FETCH_SIZE / WRITE_SIZE separately: TCC slot limits; one SQ pass with the wait split and the LDS counters).
# The profiled program (tools/shape_case.py) stands directly behind `--`.  tools/prof_shapes_summary.py folds the CSVs.
out=$1; shift
cases=("$@")
if [ ${#cases[@]} -eq 0 ]; then
  cases=("bench_nchw:256,3,50176" "per_tensor_flat:1,1,38535168" "rows_16k_x2048:1,16384,2048" "hwio_3x3x2048x2048:9,2048,2048"
         "rows_8k_x4100:1,8192,4100" "rows_8k_x4099:1,8192,4099" "rows_64k_x512:1,65536,512" "rows_1m_x32:1,1048576,32"
         "nhwc_c3:12845056,3,1" "nhwc_c64:602112,64,1" "nhwc_c256:150528,256,1" "col_6144:6144,6144,1" "inner8:2048,2048,8"
         # off the friendly grid: 7x7 planes, rows of 300 / 1001 / 5000, 100 and 1001 columns
         "planes_7x7:256,2048,49" "rows_112k_x300:1,114688,300" "rows_32k_x1001:1,32768,1001" "rows_8k_x5000:1,8192,5000"
         "cols_100:327680,100,1" "cols_1001:32768,1001,1")
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for c in "${cases[@]}"; do
  name=${c%%:*}; d=${c##*:}; IFS=, read o g i <<< "$d"
  mkdir -p $out/$name
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name/stats -- python3 tools/shape_case.py $o $g $i --iters 20 > $out/$name/stats.log 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/$name/fetch -- python3 tools/shape_case.py $o $g $i --iters 6 > $out/$name/fetch.log 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/$name/write -- python3 tools/shape_case.py $o $g $i --iters 6 > $out/$name/write.log 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $out/$name/sq -- python3 tools/shape_case.py $o $g $i --iters 6 > $out/$name/sq.log 2>&1 || exit 1
  echo "profiled $name"
done
python3 tools/prof_shapes_summary.py $out "${cases[@]}" > $out/summary.json && echo "summary written"
python3 tools/prof_shapes_summary.py --table $out/summary.json > $out/table.md && cat $out/table.md
