#!/bin/bash
cp ab/libvali_hip_B.so vali_amd/libvali_hip.so
echo "== tests (B)"; timeout 900 python -m pytest tests/test_gpu_rotate.py tests/test_gpu_rotate_staged.py tests/test_gpu_tuning.py -x -q 2>&1 | tail -3
timeout 200 python tools/stress_rotate.py 81 40 2>&1 | tail -1
echo "== A (per-pass branches, 64x64 for Y) vs B (straight-line staging in two sizes, 64x128 for one-channel 8-bit, 64x64 for 16-bit)"
TOOL=tools/rotate_any.py tools/exp/ab.sh "A B" "RGB 1920 1080 30" "RGB 1920 1080 10" "RGB 1920 1080 2" "RGB 3840 2160 30" "Y 1920 1080 30" "YUV420 1920 1080 30" "YUV444 1920 1080 10" "YUV444_10bit 1920 1080 30" "RGB_32F 1920 1080 30" 2>&1 | grep -v amdgpu.ids
cp ab/libvali_hip_B.so vali_amd/libvali_hip.so
echo "== forms for Y (B): 4 = 64x64, 5 = 64x128"
for f in 4 5 3; do for c in "Y 1920 1080 30" "YUV420 1920 1080 45" "Y 3840 2160 30"; do echo -n "form $f $c: "; VALI_ROTATE_AFFINE=$f python tools/rotate_any.py $c 2>&1 | tail -1; done; done
