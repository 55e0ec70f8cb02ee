cp ab/libvali_hip_H1.so vali_amd/libvali_hip.so
for rep in 1 2; do for r in 26 31 34 38 42 0; do echo -n "2160p rps=$((r-10)): "; VALI_RESIZE_NO_SEPARABLE=$r python tools/resize_one.py lanczos 3840 2160 1936 1088 2>&1 | grep -v amdgpu; done; done
for r in 18 21 26 34 0; do echo -n "1080p rps=$((r-10)): "; VALI_RESIZE_NO_SEPARABLE=$r python tools/resize_one.py lanczos 1920 1080 1278 718 2>&1 | grep -v amdgpu; done
