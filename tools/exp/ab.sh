#!/bin/bash
# A/B of whole libraries on ONE GPU box (boxes differ by +-5 %, so numbers from two gpurun calls do not compare):
#   build variant A, `cp vali_amd/libvali_hip.so ab/libvali_hip_A.so`, build variant B likewise (ab/ is git-ignored, travels with gpurun),
#   then  gpurun -- 'tools/exp/ab.sh "A B" "lanczos 3840 2160 1936 1088" "lanczos 2560 1440 1920 1080" ...'
# every case runs under every variant, interleaved, three times (tools/resize_one.py arguments; edit TOOL for the UD / rotate one-shots).
TOOL=${TOOL:-tools/resize_one.py}
variants=$1; shift
for rep in 1 2 3; do
  for c in "$@"; do
    for v in $variants; do
      cp ab/libvali_hip_$v.so vali_amd/libvali_hip.so
      echo -n "$v $c: "; python $TOOL $c
    done
  done
done
