#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + the two HBM counter passes for
# the headline bench command.  Outputs under gpurun_out/prof_$TAG/ (csv); summarise with
# tools/summarize_profile.py and commit the summary under profiles/.
#   usage: tools/profile.sh TAG [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --cpu-seconds 0 --ingest-seconds 0 --no-parity --no-secondary $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- \
    python $REPO/bench.py $ARGS > $OUT/trace_bench.log 2>&1
# PMC passes are separate runs (never combined with tracing)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- \
    python $REPO/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --ingest-seconds 0 --no-parity --no-secondary $* > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- \
    python $REPO/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --ingest-seconds 0 --no-parity --no-secondary $* > $OUT/pmc_write.log 2>&1
# the un-profiled number for the same command, for the record
python $REPO/bench.py $ARGS > $OUT/bench_unprofiled.log 2>&1
find $OUT -name '*.csv' | head -20
