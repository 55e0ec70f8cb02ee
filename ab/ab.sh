#!/bin/bash
cases=("lanczos 3840 2160 1920 1080 RGB_32F" "lanczos 3840 2160 1936 1088 RGB_32F" "lanczos 1920 1080 1280 720 RGB_32F_PLANAR")
for rep in 1 2 3; do
for c in "${cases[@]}"; do
  for v in head prev new d3; do
    cp ab/libvali_hip_$v.so vali_amd/libvali_hip.so
    echo -n "$v $c: "; python tools/resize_one.py $c
  done
done
done
cp ab/libvali_hip_new.so vali_amd/libvali_hip.so
