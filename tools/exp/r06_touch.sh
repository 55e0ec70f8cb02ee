#!/bin/bash
# L2 touch-prefetch VALI_TOUCH_ROWS rows ahead of the register prefetch in k_resize_rows_rgb / k_resize_rows_reg: T0 off, T6, T12
cp vali_amd/libvali_hip.so ab/libvali_hip_SAVE.so
TOOL=tools/resize_any.py tools/exp/ab.sh "T0 T6 T12" "lanczos 1280 720 1600 900 RGB" "lanczos 1280 720 1600 900" 2>&1 | grep -v amdgpu.ids
echo "== resident frames"
TOOL=tools/resize_one.py tools/exp/ab.sh "T0 T12" "lanczos 1280 720 1600 900 RGB" 2>&1 | grep -v amdgpu.ids
cp ab/libvali_hip_SAVE.so vali_amd/libvali_hip.so
