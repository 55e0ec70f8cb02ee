cp ab/libvali_hip_T26.so vali_amd/libvali_hip.so
timeout 1200 python -m pytest tests/test_gpu_resize.py tests/test_gpu_random_geometry.py tests/test_gpu_edge_geometry.py -x -q -m gpu 2>&1 | tail -3
tools/exp/ab.sh "X6 T26 T35 T45" "lanczos 3840 2160 1936 1088" "lanczos 1920 1080 1278 718" "lanczos 1920 1080 1277 719 RGB" 2>&1 | grep -v amdgpu.ids
