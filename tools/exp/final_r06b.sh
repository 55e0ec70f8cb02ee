set -x
export VALI_PROFILE_TAG=r06
mkdir -p gpurun_out
python tools/profile_secondary.py > gpurun_out/r06_profile_secondary.log 2>&1
cp gpurun_out/r06_secondary_traffic.json profiles/
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
python bench.py --verbose > gpurun_out/r06_bench_line_verbose.json 2> gpurun_out/r06_bench_line_verbose.err
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_tap_tables.py tests/test_gpu_perf_floor.py -q 2>&1 | tail -5 > gpurun_out/r06_suite_b.log
timeout 200 python tools/stress_ud.py 71 60 2>&1 | tail -1 > gpurun_out/r06_stress_b.log
timeout 200 python tools/stress_convert.py 72 60 2>&1 | tail -1 >> gpurun_out/r06_stress_b.log
cat gpurun_out/r06_suite_b.log gpurun_out/r06_stress_b.log; wc -c gpurun_out/r06_bench_line.json
