cp ab/libvali_hip_D2.so vali_amd/libvali_hip.so
timeout 1500 python -m pytest tests/test_gpu_resize.py tests/test_gpu_random_geometry.py tests/test_gpu_edge_geometry.py tests/test_gpu_gather_paths.py tests/test_gpu_tuning.py -x -q -m gpu 2>&1 | tail -3
python tools/stress_resize.py 2>&1 | tail -2
tools/exp/ab.sh "FIN D2" "lanczos 3840 2160 1936 1088" "lanczos 1920 1080 1278 718" "lanczos 1920 1080 1277 719 RGB" "lanczos 3840 2160 1920 1088" "lanczos 1920 1080 1280 720" 2>&1 | grep -v amdgpu.ids
