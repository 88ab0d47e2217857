#!/bin/bash
out=$1
for a in 4 6 9; do
  export LQ_HIP_LIB=$GRAFT_REPO_ROOT/learned_quantization_amd/csrc/liblq_hip_a$a.so
  bash tools/prof_batch_quick.sh $out/a$a imagenette:channelwise resnet50:channelwise -- --companion-only > $out.a$a.txt 2>&1 || { tail -n 5 $out.a$a.txt; exit 1; }
  echo "ring $a:"; grep "traverse<[89]>\|abi_oihw" $out.a$a.txt | cut -c1-140
done
