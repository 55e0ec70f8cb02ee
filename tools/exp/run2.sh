cp ab/libvali_hip_G1.so vali_amd/libvali_hip.so
timeout 1500 python -m pytest tests/test_gpu_resize.py tests/test_gpu_random_geometry.py tests/test_gpu_edge_geometry.py tests/test_gpu_ud.py -x -q -m gpu 2>&1 | tail -3
python tools/stress_resize.py 2>&1 | tail -2
tools/exp/ab.sh "D2 G1" "lanczos 1280 720 1920 1080" "lanczos 1280 720 1600 900" "lanczos 1280 720 1920 1080 P10" "cubic 1280 720 1920 1080" 2>&1 | grep -v amdgpu.ids
