#!/bin/bash
# A = all consumer stores non-temporal (round 5), B = partial head / tail sectors plain (VERDICT r05 #5b)
TOOL=tools/resize_any.py tools/exp/ab.sh "A B" "lanczos 3840 2160 1936 1088" "lanczos 1920 1080 1278 718" "lanczos 1920 1080 1277 719 RGB" 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
for v in A B; do
  cp $GRAFT_REPO_ROOT/ab/libvali_hip_$v.so $GRAFT_REPO_ROOT/vali_amd/libvali_hip.so
  for c in "lanczos 3840 2160 1936 1088" "lanczos 1920 1080 1278 718" "lanczos 1920 1080 1277 719 RGB"; do
    tag=$(echo $v $c | tr ' ' '_')
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_split/$tag/$ctr -o p -- python $GRAFT_REPO_ROOT/tools/resize_one.py $c > /dev/null 2>&1
    done
  done
done
python3 - <<PY
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_split"
for tag in sorted(os.listdir(root)):
    best = collections.defaultdict(float)
    for f in glob.glob(f"{root}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cols_ws" in r["Kernel_Name"]:
                best[r["Counter_Name"]] = max(best[r["Counter_Name"]], float(r["Counter_Value"]))
    print(tag, {k: round(v * 1024 / 64) for k, v in best.items()}, "bytes per frame (FETCH x 2 for HBM reads)")
PY
cp $GRAFT_REPO_ROOT/ab/libvali_hip_B.so $GRAFT_REPO_ROOT/vali_amd/libvali_hip.so
