#!/usr/bin/env python3
"""Summarise gpurun_out/prof_TAG (tools/profile.sh) into profiles/TAG_<name>.md."""
import csv
import glob
import json
import sys
from pathlib import Path

tag, name = sys.argv[1], sys.argv[2]
root = Path(__file__).resolve().parent.parent
src = root / "gpurun_out" / f"prof_{tag}"
out = root / "profiles" / f"{tag}_{name}.md"


def rows(pattern):
    for f in glob.glob(str(src / pattern), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)


lines = [f"# rocprofv3 summary {tag} / {name}", ""]
bench = [l for l in (src / "trace_bench.log").read_text().splitlines() if l.startswith("{")]
unprof = [l for l in (src / "bench_unprofiled.log").read_text().splitlines() if l.startswith("{")]
if unprof:
    j = json.loads(unprof[-1])
    lines += ["Un-profiled bench line for the same command:", "", "```json", json.dumps(j), "```", ""]
lines += ["## `rocprofv3 --kernel-trace --stats` (kernel_stats.csv)", "",
          "| kernel | calls | total ns | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|---|"]
for r in rows("trace/**/*kernel_stats.csv"):
    lines.append(f"| `{r['Name']}` | {r['Calls']} | {r['TotalDurationNs']} | {r['AverageNs']} | "
                 f"{r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
lines += [""]
per = {}
for cname, pat in (("FETCH_SIZE", "pmc_fetch/**/*counter_collection.csv"),
                   ("WRITE_SIZE", "pmc_write/**/*counter_collection.csv")):
    vals = [float(r["Counter_Value"]) for r in rows(pat)
            if r["Counter_Name"] == cname and "k_nv12_rgb8" in r["Kernel_Name"]]
    if vals:
        per[cname] = sum(vals) / len(vals)
if per:
    fetch = per.get("FETCH_SIZE", 0) * 1024 * 2   # gfx950: FETCH_SIZE reads half (MI355X_MICROARCH.md HBM)
    write = per.get("WRITE_SIZE", 0) * 1024
    lines += ["## HBM traffic per launch (separate `--pmc` passes)", "",
              f"* FETCH_SIZE avg = {per.get('FETCH_SIZE', 0):.1f} KiB raw -> x1024 x2 (gfx950 wide-read "
              f"under-count, MI355X_MICROARCH.md section HBM) = **{fetch:.4g} B**",
              f"* WRITE_SIZE avg = {per.get('WRITE_SIZE', 0):.1f} KiB raw -> x1024 = **{write:.4g} B**",
              f"* total = **{fetch + write:.5g} B per launch**", ""]
    if unprof:
        cfg = json.loads(unprof[-1])["config"]
        alg = cfg["bytes_per_frame"] * cfg["frames_per_gpu"]
        lines += [f"* algorithmic bytes per launch = {cfg['bytes_per_frame']} x {cfg['frames_per_gpu']} "
                  f"= {alg} B -> traffic / algorithmic = {(fetch + write) / alg:.4f}", ""]
if per and unprof:
    cfg = json.loads(unprof[-1])["config"]
    (root / "profiles" / f"{tag}_{name}.json").write_text(json.dumps({
        "workload": cfg["workload"], "frames_per_gpu": cfg["frames_per_gpu"],
        "bytes_per_frame": cfg["bytes_per_frame"], "fetch_bytes_per_launch": fetch,
        "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), profiles/{tag}_{name}.md; "
                  "FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 wide-read under-count)"}, indent=1) + "\n")
out.write_text("\n".join(lines) + "\n")
print(out.read_text())
