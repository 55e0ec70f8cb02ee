#!/bin/bash
mkdir -p gpurun_out
cp ab/libvali_hip_B.so vali_amd/libvali_hip.so
echo "== tests (library B)"; timeout 1500 python -m pytest tests/test_gpu_rotate.py tests/test_gpu_tuning.py tests/test_gpu_tap_tables.py tests/test_gpu_surface.py -x -q 2>&1 | tail -8
echo "== stress default"; timeout 200 python tools/stress_rotate.py 51 40 2>&1 | tail -1
echo "== A (passes unrolled) vs B (passes rolled), default forms"
TOOL=tools/rotate_any.py tools/exp/ab.sh "A B" "RGB 1920 1080 30" "RGB 1920 1080 10" "RGB 3840 2160 30" "Y 1920 1080 30" "YUV420 1920 1080 30" "YUV444_10bit 1920 1080 30" 2>&1 | grep -v amdgpu.ids
cp ab/libvali_hip_B.so vali_amd/libvali_hip.so
echo "== single calls, 1080p and 640x360, forms 6 (32x32) 3 (64x32) 4 (64x64) 1 (gather)"
for f in 6 3 4 1 0; do for c in "RGB 1920 1080 30 single" "RGB 640 360 30 single" "YUV420 1920 1080 30 single"; do echo -n "form $f $c: "; VALI_ROTATE_AFFINE=$f python tools/rotate_any.py $c 2>&1 | tail -1; done; done
echo "== small batches: 640x360 batch 64"
for f in 6 3 4 0; do echo -n "form $f RGB 640 360: "; VALI_ROTATE_AFFINE=$f python tools/rotate_any.py RGB 640 360 30 2>&1 | tail -1; done
echo "== PMC profile of the default form"
bash tools/prof_pmc.sh r06_affine "python $GRAFT_REPO_ROOT/tools/rotate_any.py RGB 1920 1080 30" > gpurun_out/prof_r06_affine.txt 2>&1
tail -45 gpurun_out/prof_r06_affine.txt
