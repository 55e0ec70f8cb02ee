cp ab/libvali_hip_N5.so vali_amd/libvali_hip.so
for c in "lanczos 1920 1080 1278 718" "lanczos 1920 1080 1277 719 RGB" "lanczos 3840 2160 1936 1088" "lanczos 1366 768 854 480"; do
for r in 0 13 14 16 18 21 26; do echo -n "$c rps=$((r-10)): "; VALI_RESIZE_NO_SEPARABLE=$r python tools/resize_one.py $c 2>&1 | grep -v amdgpu; done; done
