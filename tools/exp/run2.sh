cp ab/libvali_hip_U6.so vali_amd/libvali_hip.so
timeout 1200 python -m pytest tests/test_gpu_resize.py tests/test_gpu_random_geometry.py tests/test_gpu_edge_geometry.py tests/test_gpu_gather_paths.py -x -q -m gpu 2>&1 | tail -3
tools/exp/ab.sh "W0 T6 U6" "lanczos 3840 2160 1936 1088" "lanczos 1920 1080 1278 718" "lanczos 1920 1080 1277 719 RGB" "cubic 3840 2160 1936 1088" "lanczos 3840 2160 1920 1088" 2>&1 | grep -v amdgpu.ids
cp ab/libvali_hip_U6.so vali_amd/libvali_hip.so; bash tools/prof_pmc.sh u6 "python /root/repo/tools/resize_one.py lanczos 3840 2160 1936 1088" > gpurun_out/prof_u6.txt 2>&1
