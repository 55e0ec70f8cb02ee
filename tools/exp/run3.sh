timeout 900 python -m pytest tests/test_gpu_resize.py -x -q -m gpu -k "three_to_two or growing or rows_per_wave" 2>&1 | tail -3
timeout 300 python tools/stress_resize.py 17 45 2>&1 | tail -3
bash tools/exp/ab.sh "T1 T4" "lanczos 1280 720 1600 900 RGB" "lanczos 1280 720 1920 1200 RGB" "lanczos 960 540 1920 1080 RGB" 2>&1 | grep -v amdgpu.ids; cp ab/libvali_hip_T4.so vali_amd/libvali_hip.so
