#!/usr/bin/env python3
"""HBM traffic of the secondary kernels from rocprofv3 PMC passes (run on the GPU box through gpurun).

For each secondary config of tools/bench_configs.py: one `--kernel-trace --stats` run and two counter runs
(FETCH_SIZE, WRITE_SIZE -- they do not fit one pass; never combined with tracing).  Per kernel the BATCH
launch is the dispatch with the largest counter value (the same kernel is also launched on single frames).
  HBM read  = FETCH_SIZE [KiB] x 1024 x 2   (gfx950 tallies the 128-byte requests of wide coalesced reads at
                                             64 B: MI355X_MICROARCH.md "HBM"; calibrated on the headline kernel,
                                             whose corrected sum equals its algorithmic bytes to 4 digits)
  HBM write = WRITE_SIZE [KiB] x 1024
Writes gpurun_out/<TAG>_secondary_traffic.json (TAG = $VALI_PROFILE_TAG, default r03) (copied to profiles/ and read by bench_configs.roofline()) and a
markdown table.
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parent.parent))
TAG = os.environ.get("VALI_PROFILE_TAG", "r06")
OUT = ROOT / "gpurun_out" / f"prof_{TAG}_secondary"
# config key -> (bench_configs function, kernel-name substring, frames per launch)
KEYS = {
    "hl1080": ("hl1080", "k_nv12_rgb8", 1024),
    "cfg3": ("cfg3", "k_resize_pointk<", 64),
    "interp_bilinear": ("interp", "k_resize<", 64),
    "interp_lanczos": ("interp", "k_resize_cols_x2<", 64),        # 2160p -> 1920x1088: exactly 2:1 along x
    "interp_lanczos_1936": ("interp", "k_resize_cols_ws<unsigned char, 12, 6, 3, 4>", 64),   # 2160p -> 1936x1088: the general columns-first form
    "interp_lanczos_1278": ("interp", "k_resize_cols_ws<unsigned char, 12, 6, 4, 5>", 64),   # NV12 1080p -> 1278x718: 4 slots, wide tiles
    "interp_lanczos_rgb_1277": ("interp", "k_resize_cols_ws<unsigned char, 3, 6, 4, 5>", 64),  # packed RGB 1080p -> 1277x719
    "interp_lanczos_720p": ("interp", "k_resize_cols_x32<", 64),  # 1080p -> 720p: 3:2 both ways
    "cfg4_ud": ("cfg4", "k_ud_half<", 64),
    "cfg4_rot": ("cfg4", "k_rotate_tile", 64),
    "cfg4_fused": ("cfg4", "k_ud_half_t<", 64),
    "udgen_1280x720": ("udgen", "k_ud_32<", 64),     # 1080p -> 720p: exactly 3:2
    "udgen_640x384": ("udgen", "k_ud_nv12<", 64),    # the any-ratio kernel
    "udgen_1920x1080": ("udgen", "k_ud_lean<", 64),   # unchanged size: colour conversion with interpolated chroma
    "udplanar_8bit": ("udplanar", "k_resize_up2<unsigned char", 64),   # YUV420 -> YUV444 1080p: one launch (chroma doubled, luma copied)
    "udplanar_16bit": ("udplanar", "k_resize_up2<unsigned short", 64),
    "upscale_1920x1080": ("upscale", "k_resize_rows_x23<", 64),        # 720p -> 1080p Lanczos (3:2 both ways)
    "upscale_1600x900": ("upscale", "k_resize_rows_reg<", 64),         # 720p -> 1600x900 (general growing planes, register form)
    "upscale_1920x1080_rgb": ("upscale", "k_resize_rows_x23_rgb<", 64),  # packed RGB 720p -> 1080p: the 3:2 form, three channels
    "upscale_1600x900_rgb": ("upscale", "k_resize_rows_rgb<", 64),     # packed RGB 720p -> 1600x900: the register form, three channels
    "upscale_1920x1080_rgb_32f": ("upscale", "k_resize_taps<", 64),    # RGB_32F 720p -> 1080p: rows-first gather kernel
    "upscale_1600x900_p10": ("upscale", "k_resize_rows<", 64),         # P10 720p -> 1600x900: LDS-staged rows form
    "affine_rgb_30": ("affine", "k_rotate_affine_lds", 64, min),      # RGB 1080p by 30 degrees about the origin: the LDS-staged form (round 6)
    "affine_rgb_30_centre": ("affine", "k_rotate_affine_lds", 64, max),  # ... about the centre of the frame: more of the destination inside the source
}


def run(cmd, log):
    with open(log, "w") as fh:
        subprocess.run(cmd, stdout=fh, stderr=subprocess.STDOUT, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))


def collect(cfg):
    d = OUT / cfg
    d.mkdir(parents=True, exist_ok=True)
    cmd = [sys.executable, str(ROOT / "tools" / "bench_configs.py"), cfg]
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", str(d / "trace"), "-o", "t", "--"] + cmd, d / "trace.log")
    run(["rocprofv3", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", str(d / "fetch"), "-o", "p", "--"] + cmd, d / "fetch.log")
    run(["rocprofv3", "--pmc", "WRITE_SIZE", "--output-format", "csv", "-d", str(d / "write"), "-o", "p", "--"] + cmd, d / "write.log")
    run(cmd, d / "unprofiled.log")
    pmc = defaultdict(lambda: defaultdict(list))
    for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        for f in glob.glob(str(d / sub / "**" / "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == name:
                    pmc[r["Kernel_Name"]][name].append(float(r["Counter_Value"]))
    stats = {}
    for f in glob.glob(str(d / "trace" / "**" / "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            stats[r["Name"]] = r
    return pmc, stats


def kernel_names():
    """--names: which kernels does every config dispatch?  One quick kernel-trace run per config (VALI_BENCH_QUICK: small
    sets, two launches per entry); prints {config: [kernel names]} as one JSON line.  tests/test_gpu_bench.py holds the
    committed traffic file against it: a profile of a kernel the library no longer launches fails a test."""
    out = {}
    for cfg in sorted({spec[0] for spec in KEYS.values()}):
        d = OUT / ("names_" + cfg)
        d.mkdir(parents=True, exist_ok=True)
        cmd = [sys.executable, str(ROOT / "tools" / "bench_configs.py"), cfg]
        with open(d / "trace.log", "w") as fh:
            subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", str(d / "trace"), "-o", "t", "--"] + cmd,
                           stdout=fh, stderr=subprocess.STDOUT, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", VALI_BENCH_QUICK="1"))
        seen = set()
        for f in glob.glob(str(d / "trace" / "**" / "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                seen.add(r["Name"].replace("void vali::", "").split("(")[0])
        out[cfg] = sorted(seen)
    print(json.dumps(out))


def traffic_of(pmc, spec):
    """HBM bytes per launch of one KEYS entry from the per-kernel counter lists of its config (see the module docstring)"""
    needle = spec[1]
    pick = spec[3] if len(spec) > 3 else max
    names = [k for k in pmc if needle in k]
    if not names:
        return None
    k = max(names, key=lambda n: max(pmc[n]["FETCH_SIZE"], default=0))
    big = lambda vals: [x for x in vals if x > 0.25 * max(vals, default=0)]   # noqa: E731 (warm-up / single-surface launches aside)
    return k, pick(big(pmc[k]["FETCH_SIZE"]), default=0) * 1024 * 2, pick(big(pmc[k]["WRITE_SIZE"]), default=0) * 1024


def check(cfgs):
    """--check cfg[,cfg]: the two PMC passes of these configs in quick mode (one small set, two launches per entry), HBM bytes per
    launch of every KEYS entry of theirs as one JSON line -- tests/test_gpu_bench.py holds the committed traffic file against it: a
    kernel whose traffic changed since the profile was taken fails a test."""
    out = {}
    for cfg in cfgs:
        d = OUT / ("check_" + cfg)
        d.mkdir(parents=True, exist_ok=True)
        cmd = [sys.executable, str(ROOT / "tools" / "bench_configs.py"), cfg]
        env = dict(os.environ, TMPDIR="/tmp", VALI_BENCH_QUICK="1")
        pmc = defaultdict(lambda: defaultdict(list))
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            with open(d / (name + ".log"), "w") as fh:
                subprocess.run(["rocprofv3", "--pmc", name, "--output-format", "csv", "-d", str(d / name), "-o", "p", "--"] + cmd, stdout=fh,
                               stderr=subprocess.STDOUT, cwd="/tmp", env=env)
            for f in glob.glob(str(d / name / "**" / "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == name:
                        pmc[r["Kernel_Name"]][name].append(float(r["Counter_Value"]))
        for key, spec in KEYS.items():
            if spec[0] == cfg:
                t = traffic_of(pmc, spec)
                if t:
                    out[key] = {"kernel": t[0].replace("void vali::", "").split("(")[0], "hbm_bytes_per_launch": t[1] + t[2]}
    print(json.dumps(out))


def main():
    if "--names" in sys.argv:
        return kernel_names()
    if "--check" in sys.argv:
        return check(sys.argv[sys.argv.index("--check") + 1].split(","))
    done, table, result = {}, [], {}
    for key, spec in KEYS.items():
        cfg, needle, frames = spec[:3]
        pick = spec[3] if len(spec) > 3 else max   # which of the kernel's launches: the largest dispatch unless said otherwise
        if cfg not in done:
            done[cfg] = collect(cfg)
        pmc, stats = done[cfg]
        names = [k for k in pmc if needle in k]
        if not names:
            continue
        k = max(names, key=lambda n: max(pmc[n]["FETCH_SIZE"], default=0))
        big = lambda vals: [x for x in vals if x > 0.25 * max(vals, default=0)]   # (warm-up / single-surface launches aside)
        rd = pick(big(pmc[k]["FETCH_SIZE"]), default=0) * 1024 * 2
        wr = pick(big(pmc[k]["WRITE_SIZE"]), default=0) * 1024
        st = stats.get(k, {})
        result[key] = {"kernel": k.replace("void vali::", "").split("(")[0], "frames": frames,
                       "hbm_read_bytes_per_launch": rd, "hbm_written_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                       "max_ns": float(st.get("MaxNs", 0) or 0), "calls": int(st.get("Calls", 0) or 0),
                       "source": f"profiles/{TAG}_secondary_traffic.json (rocprofv3 --pmc FETCH_SIZE x1024 x2 + WRITE_SIZE x1024, "
                                 "separate passes, batch launch = largest dispatch)"}
        table.append(f"| {key} | `{result[key]['kernel']}` | {frames} | {rd:.5g} | {wr:.5g} | {rd + wr:.5g} | {st.get('MaxNs', '')} |")
    if "cfg4_ud" in result and "cfg4_rot" in result:
        a, b = result["cfg4_ud"], result["cfg4_rot"]
        result["cfg4_chain"] = {"kernel": "k_ud_half + k_rotate_tile", "frames": 64,
                                "hbm_bytes_per_launch": a["hbm_bytes_per_launch"] + b["hbm_bytes_per_launch"], "source": a["source"]}
    (ROOT / "gpurun_out" / f"{TAG}_secondary_traffic.json").write_text(json.dumps(result, indent=1) + "\n")
    # the kernel-stats table of every config's trace run, as rocprofv3 wrote it: frac can be recomputed from these alone
    import shutil
    for cfg in done:
        for f in glob.glob(str(OUT / cfg / "trace" / "**" / "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, ROOT / "gpurun_out" / f"{TAG}_{cfg}_kernel_stats.csv")
    md = [f"# {TAG}: HBM traffic of the secondary kernels (rocprofv3 PMC, tools/profile_secondary.py)", "",
          "Every config cycles >= 1.5 GiB of distinct surface sets per timed loop (tools/bench_configs.py); counters are per launch "
          "of one 64-frame set.", "",
          "| config | kernel | frames per launch | HBM read B | HBM written B | sum | longest launch ns |", "|---|---|---|---|---|---|---|"] + table
    md += ["", "un-profiled bench lines of the same box:", "", "```"]
    for cfg in done:
        md += [l for l in (OUT / cfg / "unprofiled.log").read_text().splitlines() if l.startswith("{")]
    md += ["```", ""]
    (ROOT / "gpurun_out" / f"{TAG}_secondary_traffic.md").write_text("\n".join(md))
    print("\n".join(md)[:5000])


if __name__ == "__main__":
    main()
