for v in R6 S1; do
cp ab/libvali_hip_$v.so vali_amd/libvali_hip.so
bash tools/prof_pmc.sh x23$v "python /root/repo/tools/resize_one.py lanczos 1280 720 1920 1080 NV12" > gpurun_out/prof_x23$v.txt 2>&1
echo "== $v"; grep -E "per launch" gpurun_out/prof_x23$v.txt | awk '{print $1, $NF}' | grep -E "ACTIVE_INST_(ANY|VALU|SCA|MISC|VMEM)|INSTS_VALU|GRBM|FETCH|WRITE"; grep "^\"void" gpurun_out/prof_x23$v.txt | cut -c1-150
done
cp ab/libvali_hip_S1.so vali_amd/libvali_hip.so
