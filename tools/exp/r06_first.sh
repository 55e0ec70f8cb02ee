#!/bin/bash
# round 6, first GPU call: LDS micro-benchmark, the LDS-staged rotation (parity + forms), A/B of the two round-5 slow-downs
mkdir -p gpurun_out
cp vali_amd/libvali_hip.so ab/libvali_hip_NEW.so
( tools/exp/lds_unaligned > gpurun_out/lds_unaligned.txt 2>&1 ) 
echo "== rotate tests"; timeout 900 python -m pytest tests/test_gpu_rotate.py -x -q 2>&1 | tail -5
echo "== stress default"; timeout 200 python tools/stress_rotate.py 11 45 2>&1 | tail -3
for f in 2 3 4 5; do echo "== stress form $f"; VALI_ROTATE_AFFINE=$f timeout 200 python tools/stress_rotate.py $((20+f)) 20 2>&1 | tail -2; done
echo "== timing (bench_configs.affine, rotating sets)"
for rep in 1 2; do for f in 1 0 2 3 4 5; do echo -n "form $f: "; VALI_ROTATE_AFFINE=$f python -c "
import sys; sys.path.insert(0,'tools')
import bench_configs as bc
r=bc.affine(); print(r['us_per_frame'], r['roofline']['frac'])" 2>&1 | tail -1; done; done
echo "== other angles / formats (rotate_any.py)"
for f in 1 0 2 3; do for c in "RGB 1920 1080 45" "RGB 1920 1080 10" "RGB 3840 2160 30" "Y 1920 1080 30" "YUV420 1920 1080 30" "RGB_32F 1920 1080 30" "YUV444_10bit 1920 1080 30"; do echo -n "form $f $c: "; VALI_ROTATE_AFFINE=$f python tools/rotate_any.py $c 2>&1 | tail -1; done; done
echo "== A/B k_ud_lean (HEAD vs PRE = HEAD with k_ud_lean's unaligned arms plain again)"
TOOL=tools/ud_one.py tools/exp/ab.sh "HEAD PRE" "1920 1080 1920 1080 RGB" "1920 1080 1920 1080 RGB_PLANAR" "1918 1078 1918 1078 RGB"
cp ab/libvali_hip_HEAD.so vali_amd/libvali_hip.so
echo "== A/B planar UD: the round-4 tree vs HEAD (round-5 library)"
for rep in 1 2 3; do
  for c in "1920 1080 1920 1080" "1920 1080 1920 1080 10"; do
    echo -n "r04 $c: "; (cd ab/r04 && python tools/udplanar_one.py $c | tail -1)
    echo -n "r05 $c: "; python tools/udplanar_one.py $c | tail -1
  done
  echo -n "r04 udlean RGB: "; (cd ab/r04 && python tools/ud_one.py 1920 1080 1920 1080 RGB | tail -1)
  echo -n "r05 udlean RGB: "; python tools/ud_one.py 1920 1080 1920 1080 RGB | tail -1
done
cp ab/libvali_hip_NEW.so vali_amd/libvali_hip.so
