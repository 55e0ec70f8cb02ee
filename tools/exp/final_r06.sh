set -x
export VALI_PROFILE_TAG=r06
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r06_suite.log
python tools/profile_secondary.py > gpurun_out/r06_profile_secondary.log 2>&1
cp gpurun_out/r06_secondary_traffic.json profiles/
bash tools/profile.sh r06 > gpurun_out/r06_profile.log 2>&1
python tools/summarize_profile.py r06 nv12_rgb_2160p_batch512 > gpurun_out/r06_summary.log 2>&1
cp gpurun_out/r06_nv12_rgb_2160p_batch512.json profiles/ 2>/dev/null
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
python bench.py --verbose > gpurun_out/r06_bench_line_verbose.json 2> gpurun_out/r06_bench_line_verbose.err
bash tools/prof_pmc.sh r06_affine "python $GRAFT_REPO_ROOT/tools/rotate_any.py RGB 1920 1080 30" > gpurun_out/r06_affine_counters.txt 2>&1
python tools/cliffs.py > gpurun_out/r06_cliffs.txt 2>&1
timeout 300 python tools/stress_rotate.py 61 120 2>&1 | tail -1 > gpurun_out/r06_stress.log
timeout 300 python tools/stress_resize.py 62 120 2>&1 | tail -1 >> gpurun_out/r06_stress.log
cat gpurun_out/r06_suite.log gpurun_out/r06_stress.log; wc -c gpurun_out/r06_bench_line.json
