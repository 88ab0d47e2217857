#!/bin/bash
out=$1
export LQ_HIP_LIB=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc/liblq_hip_dev.so
cases() { for L in "$@"; do echo -n "r$L:1,$((33554432 / L)),$L "; done; }
one() { tag=$1; shift; envs=$1; shift; d=$out/$tag; mkdir -p $d; CASES="$*" bash tools/r02_awkward.sh $d $envs > /dev/null || exit 1; sed "s/^/$tag  /" $d/sweep.txt >> $out/win_store.txt; }
rm -f $out/win_store.txt
L1="66 68 77 84 88 100 120"
L2="130 132 150 168 200"
L3="258 260 300"
one shipped_nt      "LQ_DEV_FLAGS=0" $(cases $L1 $L2 $L3)
one shipped_dflt    "LQ_DEV_FLAGS=1" $(cases $L1 $L2 $L3)
one g421_dflt       "LQ_DEV_FLAGS=1 LQ_TUNE_WIN_GEOM=421" $(cases $L1)
one g431_dflt       "LQ_DEV_FLAGS=1 LQ_TUNE_WIN_GEOM=431" $(cases $L2)
one g531_dflt       "LQ_DEV_FLAGS=1 LQ_TUNE_WIN_GEOM=531" $(cases $L3)
cut -c1-20,150-230 $out/win_store.txt
