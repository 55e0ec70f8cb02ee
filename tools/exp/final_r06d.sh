set -x
export VALI_PROFILE_TAG=r06
mkdir -p gpurun_out
python tools/profile_secondary.py > gpurun_out/r06_profile_secondary.log 2>&1
cp gpurun_out/r06_secondary_traffic.json profiles/
bash tools/profile.sh r06 > gpurun_out/r06_profile.log 2>&1
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
python bench.py --verbose > gpurun_out/r06_bench_line_verbose.json 2> gpurun_out/r06_bench_line_verbose.err
bash tools/prof_pmc.sh r06_affine "python $GRAFT_REPO_ROOT/tools/rotate_any.py RGB 1920 1080 30" > gpurun_out/r06_affine_counters.txt 2>&1
bash tools/prof_pmc.sh r06_affine_y "python $GRAFT_REPO_ROOT/tools/rotate_any.py Y 1920 1080 30" > gpurun_out/r06_affine_y_counters.txt 2>&1
python tools/cliffs.py > gpurun_out/r06_cliffs.txt 2>&1
wc -c gpurun_out/r06_bench_line.json
