#!/bin/bash
# usage: tools/r03_win_geom.sh <outdir>   (GPU box, repo root; development library) -- K2 / K4 of short rows (outer == 1) under
# forced row-window team geometries: LQ_TUNE_WIN_GEOM = LG * 100 + V * 10 + U (teams of 2^LG lanes, V float4 per lane, U rows per
# team) against the shipped choice.  32 M elements per descriptor, rocprofv3 kernel durations (tools/r02_awkward.sh).
out=$1
export LQ_HIP_LIB=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc/liblq_hip_dev.so
one() {  # one <geom or 0> <cases>
  g=$1; shift
  d=$out/g$g; mkdir -p $d
  if [ "$g" = "0" ]; then CASES="$*" bash tools/r02_awkward.sh $d > /dev/null || exit 1
  else CASES="$*" bash tools/r02_awkward.sh $d LQ_TUNE_WIN_GEOM=$g > /dev/null || exit 1; fi
  sed "s/^/geom $g  /" $d/sweep.txt >> $out/win_geom.txt
}
rm -f $out/win_geom.txt
cases() { for L in "$@"; do echo -n "r$L:1,$((33554432 / L)),$L "; done; }
ALL="66 68 72 77 80 84 88 92 96 100 104 112 120 128 130 131 132 136 140 148 150 156 160 168 176 184 192 200 201 208 224 240 256 258 260 272 288 300 320 352 384"
one 0   $(cases $ALL)
if [ "$FINAL" = "1" ]; then cat $out/win_geom.txt | cut -c1-220; exit 0; fi      # the shipped rules only
one 421 $(cases 66 68 72 77 80 84 88 92 96 100 104 112 120 128)
one 431 $(cases 130 131 132 136 140 148 150 156 160 168 176 184 192)
one 521 $(cases 130 131 132 136 140 148 150 156 160 168 176 184 192 200 201 208 224 240 256)
one 531 $(cases 258 260 272 288 300 320 352 384)
echo "done: $(wc -l < $out/win_geom.txt) lines"
