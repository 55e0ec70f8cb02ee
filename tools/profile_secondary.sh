#!/bin/bash
# Run on the GPU box (via gpurun): per secondary config (tools/bench_configs.py cfg3 / cfg4 /
# preproc) one rocprofv3 kernel-stats run and the two HBM counter passes (PMC passes are
# separate runs, never combined with tracing).
#   usage: tools/profile_secondary.sh TAG
set -u
TAG=${1:-r01s}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for CFG in cfg3 cfg4 preproc; do
  OUT=$REPO/gpurun_out/prof_$TAG/$CFG
  mkdir -p $OUT
  CMD="python $REPO/tools/bench_configs.py $CFG"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace_bench.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
  $CMD > $OUT/bench_unprofiled.log 2>&1
done
find $REPO/gpurun_out/prof_$TAG -name '*kernel_stats.csv'
