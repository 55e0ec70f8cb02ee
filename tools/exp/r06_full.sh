#!/bin/bash
mkdir -p gpurun_out
echo "== full GPU suite"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "== bench"; timeout 900 python bench.py > gpurun_out/r06_bench_line_a.json 2> gpurun_out/r06_bench_a.err; wc -c gpurun_out/r06_bench_line_a.json; tail -2 gpurun_out/r06_bench_a.err
echo "== growing planes: B = fixed 64 rows per wave, C = balanced rows"; TOOL=tools/resize_one.py tools/exp/ab.sh "B C" "lanczos 1280 720 1600 900" "lanczos 1280 720 1600 900 RGB" "lanczos 1280 720 1600 900 P10" "lanczos 1280 720 1500 850" 2>&1 | grep -v amdgpu.ids; cp ab/libvali_hip_C.so vali_amd/libvali_hip.so
echo "== staged form with 64-row waves (VERDICT r05 #4b)"; for c in "lanczos 1280 720 1600 900" "lanczos 1280 720 1600 900 RGB"; do echo -n "ROWS=3 SEP=4 $c: "; VALI_RESIZE_ROWS=3 VALI_RESIZE_NO_SEPARABLE=4 python tools/resize_one.py $c 2>&1 | tail -1; done
echo "== secondary traffic profile"; VALI_PROFILE_TAG=r06 timeout 2400 python tools/profile_secondary.py > gpurun_out/r06_profile_secondary.log 2>&1; tail -3 gpurun_out/r06_profile_secondary.log
