cp ab/libvali_hip_X6.so vali_amd/libvali_hip.so
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
tools/exp/ab.sh "W0 V6 X6" "lanczos 3840 2160 1936 1088" "lanczos 1920 1080 1278 718" "lanczos 1920 1080 1277 719 RGB" "lanczos 1366 768 854 480" 2>&1 | grep -v amdgpu.ids
