#!/bin/bash
cp ab/libvali_hip_new.so vali_amd/libvali_hip.so
python -m pytest tests/test_gpu_resize.py tests/test_gpu_gather_paths.py -x -q -m gpu 2>&1 | tail -2
cases=("lanczos 3840 2160 1936 1088" "lanczos 3840 2160 1920 1088" "lanczos 1920 1080 1280 720" "lanczos 2560 1440 1920 1080" "lanczos 3840 2160 1936 1088 RGB" "lanczos 3840 2160 1920 1080 RGB_32F" "lanczos 1920 1080 640 384" "cubic 3840 2160 1936 1088")
for rep in 1 2; do
for c in "${cases[@]}"; do
  for v in head prev new; do
    cp ab/libvali_hip_$v.so vali_amd/libvali_hip.so
    echo -n "$v $c: "; python tools/resize_one.py $c
  done
done
done
cp ab/libvali_hip_new.so vali_amd/libvali_hip.so
