timeout 900 python -m pytest tests/test_gpu_resize.py -x -q -m gpu -k "three_to_two or growing" 2>&1 | tail -3
timeout 300 python tools/stress_resize.py 11 60 2>&1 | tail -3
bash tools/exp/ab.sh "R5 R6" "lanczos 1280 720 1920 1080 RGB" "cubic 1280 720 1920 1080 RGB" "lanczos 2560 1440 3840 2160 RGB" 2>&1 | grep -v amdgpu.ids; cp ab/libvali_hip_R6.so vali_amd/libvali_hip.so
