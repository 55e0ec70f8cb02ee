set -x
export VALI_PROFILE_TAG=r05
python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
bash tools/profile.sh r05 > gpurun_out/r05_profile.log 2>&1
python tools/summarize_profile.py r05 nv12_rgb_2160p_batch512 > gpurun_out/r05_summary.log 2>&1
python tools/profile_secondary.py > gpurun_out/r05_profile_secondary.log 2>&1
bash tools/prof_pmc.sh r05ws "python /root/repo/tools/resize_one.py lanczos 3840 2160 1936 1088" > gpurun_out/r05_ws_counters.txt 2>&1
python tools/cliffs.py > gpurun_out/r05_cliffs.txt 2>&1
bash tools/prof_pmc.sh r05rgb23 "python /root/repo/tools/resize_one.py lanczos 1280 720 1920 1080 RGB" > gpurun_out/r05_rgb23_counters.txt 2>&1
