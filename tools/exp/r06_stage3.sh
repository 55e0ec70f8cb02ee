#!/bin/bash
cp ab/libvali_hip_B.so vali_amd/libvali_hip.so
echo "== tests (B)"; timeout 900 python -m pytest tests/test_gpu_rotate.py tests/test_gpu_rotate_staged.py tests/test_gpu_tuning.py -x -q 2>&1 | tail -3
timeout 200 python tools/stress_rotate.py 91 40 2>&1 | tail -1
echo "== A (linear piece mapping) vs B (row-mapped staging for three-channel planes)"
TOOL=tools/rotate_any.py tools/exp/ab.sh "A B" "RGB 1920 1080 30" "RGB 1920 1080 10" "RGB 1920 1080 45" "RGB 3840 2160 30" "RGB 640 360 30" "RGB_32F 1920 1080 30" "Y 1920 1080 30" 2>&1 | grep -v amdgpu.ids
cp ab/libvali_hip_B.so vali_amd/libvali_hip.so
