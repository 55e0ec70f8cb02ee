tools/exp/ab.sh "P0 D6 D8" "lanczos 3840 2160 1936 1088" "lanczos 1920 1080 1278 718" 2>&1 | grep -v amdgpu.ids
for lib in D8; do cp ab/libvali_hip_$lib.so vali_amd/libvali_hip.so
for rep in 1; do for w in 20 16 12 8; do echo -n "$lib waves/CU $w: "; VALI_WAVES_PER_CU=$w python tools/resize_one.py lanczos 3840 2160 1936 1088 2>&1 | grep -v amdgpu; done; done; done
