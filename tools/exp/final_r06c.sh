set -x
export VALI_PROFILE_TAG=r06
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r06_suite.log
python tools/profile_secondary.py > gpurun_out/r06_profile_secondary.log 2>&1
cp gpurun_out/r06_secondary_traffic.json profiles/
bash tools/profile.sh r06 > gpurun_out/r06_profile.log 2>&1
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
python bench.py --verbose > gpurun_out/r06_bench_line_verbose.json 2> gpurun_out/r06_bench_line_verbose.err
python tools/cliffs.py > gpurun_out/r06_cliffs.txt 2>&1
cat gpurun_out/r06_suite.log; wc -c gpurun_out/r06_bench_line.json
