#!/bin/bash
mkdir -p gpurun_out
echo "== rotate tests"; timeout 900 python -m pytest tests/test_gpu_rotate.py tests/test_gpu_tuning.py -x -q 2>&1 | tail -3
echo "== stress default"; timeout 200 python tools/stress_rotate.py 31 30 2>&1 | tail -1
for f in 2 3 4 5; do echo "== stress form $f"; VALI_ROTATE_AFFINE=$f timeout 200 python tools/stress_rotate.py $((40+f)) 12 2>&1 | tail -1; done
echo "== timing (bench_configs.affine, rotating sets)"
for rep in 1 2; do for f in 1 0 2 3 4 5; do echo -n "form $f: "; VALI_ROTATE_AFFINE=$f python -c "
import sys; sys.path.insert(0,'tools')
import bench_configs as bc
r=bc.affine(); print(r['us_per_frame'], r['roofline']['frac'])" 2>&1 | tail -1; done; done
echo "== other angles / formats (rotate_any.py)"
for f in 1 0 2 3; do for c in "RGB 1920 1080 45" "RGB 1920 1080 10" "RGB 1920 1080 30 500 -300" "RGB 3840 2160 30" "Y 1920 1080 30" "YUV420 1920 1080 30" "RGB_32F 1920 1080 30" "YUV444_10bit 1920 1080 30"; do echo -n "form $f $c: "; VALI_ROTATE_AFFINE=$f python tools/rotate_any.py $c 2>&1 | tail -1; done; done
echo "== resize / tap tables"; timeout 1500 python -m pytest tests/test_gpu_resize.py tests/test_gpu_random_geometry.py tests/test_gpu_ud.py -x -q 2>&1 | tail -3
