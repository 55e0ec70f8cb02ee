#!/usr/bin/env python3
"""Summarise gpurun_out/prof_TAG (tools/profile_secondary.sh) into profiles/TAG_secondary_kernels.md:
per kernel the rocprofv3 average duration and the HBM bytes per launch from the PMC passes
(FETCH_SIZE x 1024 x 2 -- gfx950 wide-read under-count, MI355X_MICROARCH.md -- and WRITE_SIZE x 1024)."""
import csv
import glob
import sys
from collections import defaultdict
from pathlib import Path

tag = sys.argv[1]
root = Path(__file__).resolve().parent.parent
src = root / "gpurun_out" / f"prof_{tag}"


def rows(pattern):
    for f in glob.glob(str(src / pattern), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)


def short(name):
    return name.split("(")[0].replace("void vali::", "").replace("vali::", "")


lines = [f"# rocprofv3 {tag}: secondary kernels (tools/bench_configs.py cfg3 / cfg4 / preproc, one profile each)", "",
         "HBM bytes per launch = max over launches of the kernel within the config's run (the 64-frame batch",
         "launch; single-frame calls of the same kernel are far smaller).  FETCH_SIZE is KiB and reads half on",
         "gfx950 (x1024 x2, MI355X_MICROARCH.md), WRITE_SIZE is KiB.  Separate --pmc passes.", ""]
base = src
for cfg in ("cfg3", "cfg4", "preproc"):
    src = base / cfg
    summarize = True
    stats = {short(r["Name"]): r for r in rows("trace/**/*kernel_stats.csv")}
    pmc = defaultdict(lambda: defaultdict(list))
    for cname, pat in (("FETCH_SIZE", "pmc_fetch/**/*counter_collection.csv"),
                       ("WRITE_SIZE", "pmc_write/**/*counter_collection.csv")):
        for r in rows(pat):
            if r["Counter_Name"] == cname:
                pmc[short(r["Kernel_Name"])][cname].append(float(r["Counter_Value"]))
    # modal launch = the batch launch (single-frame calls of the same kernel are far smaller)
    lines += [f"## {cfg}", "",
             "| kernel | calls | avg ns (all launches) | max ns | HBM read B (batch launch) | HBM written B | read+written |",
             "|---|---|---|---|---|---|---|"]
    for k, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
        f = max(pmc[k]["FETCH_SIZE"], default=0) * 1024 * 2
        w = max(pmc[k]["WRITE_SIZE"], default=0) * 1024
        lines.append(f"| `{k}` | {r['Calls']} | {r['AverageNs']} | {r['MaxNs']} | {f:.4g} | {w:.4g} | {f + w:.4g} |")
    lines += ["", "## bench lines of the un-profiled run on the same box", "", "```"]
    lines += [l for l in (src / "bench_unprofiled.log").read_text().splitlines() if l.startswith("{")]
    lines += ["```", ""]
out = root / "profiles" / f"{tag}_secondary_kernels.md"
out.write_text("\n".join(lines) + "\n")
print(out.read_text()[:6000])
