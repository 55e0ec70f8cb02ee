#!/bin/bash
# usage: tools/prof_tcc.sh TAG "<command>" : L2 / fabric-side counters of every kernel of the command, one rocprofv3 --pmc pass per
# counter set (never combined with tracing), summed over the 16 TCC channels (_sum).  csv + a table under gpurun_out/tcc_TAG.
# What it is for: a kernel at HBM traffic 1.0x its algorithmic bytes that still runs below the copy ceiling -- request SIZES
# (32 / 64 / 128-byte fabric requests), partial-line writes, write-request stalls, tag stalls, L2 hit rates.
set -u
TAG=$1; CMD=$2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/tcc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_READ_sum TCC_WRITE_sum TCC_WRITEBACK_sum" \
           "TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_IB_STALL_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1 || echo "pass $i ($SET) failed" >> $OUT/failed.txt
done
python3 - <<PY
import csv, glob, collections
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "copyBuffer" in k or "fill" in k.lower():
        continue
    print(k)
    for c, v in sorted(d.items()):
        big = [x for x in v if x > 0.25 * max(v)] or v      # the batch launches (single-surface warm-ups aside)
        print("   %-36s per launch %16.0f   (%d launches)" % (c, sum(big) / len(big), len(big)))
PY
