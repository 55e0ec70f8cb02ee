"""bench.py contract on a real GPU: one JSON line with the required keys; the N>1 code path
(sharding, coefficient broadcast, barrier, max-over-ranks) with 2 ranks sharing the one GPU."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline"]


def last_json(out: str):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract(gpu):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1",
                        "--frames", "16", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = last_json(r.stdout)
    for k in REQUIRED:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["unit"] == "frames/s" and j["dtype"] == "u8"
    assert j["ranks_seen"] == 1 and j["launcher"] == "single process"
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert j["parity_vs_oracle"]["max_abs_diff_lsb"] == 0
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["verification"] == {"frames_verified_on_gpu": 16, "frames_mismatching": 0,
                                 "checksum_of_checksums": j["verification"]["checksum_of_checksums"]}
    # the default line is the compact one: every secondary measurement as one short object, weakest last,
    # the whole line short enough for a reader that keeps an 8 KB tail
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][0]
    assert len(line) < 7500, len(line)
    sec = j["secondary"]
    assert isinstance(sec, list) and all(isinstance(e, dict) and "cfg" in e for e in sec)
    roof = [e for e in sec if "frac" in e]
    assert all(set(e) == {"cfg", "kernel", "us", "B", "frac", "traffic"} for e in roof), roof
    fr = [e["frac"] for e in roof]
    assert fr == sorted(fr, reverse=True) and all(0 < f < 1.0 for f in fr) and len(fr) >= 24, fr
    cfgs = {e["cfg"] for e in sec}
    for want in ("interp lanczos NV12 1920x1080->1278x718", "interp lanczos RGB 1920x1080->1277x719", "interp lanczos NV12 3840x2160->1936x1088",
                 "interp bilinear NV12 3840x2160->1920x1088", "upscale lanczos NV12 1280x720->1920x1080", "upscale lanczos NV12 1280x720->1600x900",
                 "upscale lanczos RGB 1280x720->1920x1080", "upscale lanczos RGB 1280x720->1600x900", "upscale lanczos RGB_32F 1280x720->1920x1080",
                 "upscale lanczos P10 1280x720->1600x900", "cfg4 ud", "cfg4 rot", "cfg4 chain", "cfg4 fused"):
        assert want in cfgs, (want, sorted(cfgs))
    for fam in ("NV12->RGB 1920x1080", "cfg2", "cfg3", "udgen", "udplanar", "affine"):
        assert any(c.startswith(fam) for c in cfgs), (fam, sorted(cfgs))
    assert all(e["kernel"].startswith("k_") for e in roof if e["cfg"] != "cfg4 chain"), roof
    assert "traffic" in j["secondary_keys"] and 0.3 < j["roofline"]["frac"] < 1.0


def test_bench_verbose_line_keeps_the_full_roofline_objects(gpu):
    """--verbose: the long form profiles/rNN_bench_line.json keeps (notes, bound / peak / unit / working set of every entry)."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "8", "--cpu-seconds", "0",
                        "--ingest-seconds", "0", "--verbose"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = last_json(r.stdout)
    assert [c["config"][:4] for c in j["secondary"]] == ["NV12", "cfg2", "cfg3", "inte", "upsc", "cfg4", "udge", "udpl", "affi"]

    def fracs(o):      # every roofline entry anywhere in the line: a fraction of the HBM peak, never above it
        if isinstance(o, dict):
            if "frac" in o and "peak" in o:
                yield o["frac"]
            for v in o.values():
                yield from fracs(v)
        elif isinstance(o, list):
            for v in o:
                yield from fracs(v)
    all_fracs = list(fracs(j))
    assert len(all_fracs) >= 24 and all(0 < f < 1.0 for f in all_fracs), all_fracs
    assert {r["filter"] for r in j["secondary"][3]["results"]} == {"bilinear", "lanczos"}


def test_committed_traffic_profiles_name_the_kernels_the_library_dispatches(gpu):
    """`roofline.traffic` of the secondary lines is replayed from profiles/r05_secondary_traffic.json (PMC counters cannot
    be read inside the run).  Every kernel named there must be a kernel the CURRENT binary launches for that config --
    compared by base name with a quick rocprofv3 kernel trace of every config (tools/profile_secondary.py --names): a
    profile that has gone stale fails here instead of decorating a line with another kernel's bytes."""
    import shutil

    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on this box")
    prof = sorted((ROOT / "profiles").glob("r*_secondary_traffic.json"))[-1]
    traffic = json.loads(prof.read_text())
    sys.path.insert(0, str(ROOT / "tools"))
    import profile_secondary as ps

    r = subprocess.run([sys.executable, str(ROOT / "tools" / "profile_secondary.py"), "--names"], capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = last_json(r.stdout)
    checked = 0
    for key, entry in traffic.items():
        if key not in ps.KEYS:
            continue
        cfg = ps.KEYS[key][0]
        for base in [k.strip().split("<")[0] for k in entry["kernel"].split("+")]:
            assert any(n.split("<")[0] == base for n in seen[cfg]), (prof.name, key, base, seen[cfg])
            checked += 1
    assert checked >= 15, checked


def test_committed_traffic_profiles_still_hold_the_bytes_the_kernels_move(gpu):
    """The bytes, not only the names: the PMC passes of three configs are re-taken in quick mode and every entry
    of theirs must sit within 6 % of profiles/r*_secondary_traffic.json (a launch is one 64-frame set either way; the non-temporal-store
    commits of round 5 moved written bytes by 4-9 % between two profiles without any test noticing)."""
    import shutil

    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on this box")
    prof = sorted((ROOT / "profiles").glob("r*_secondary_traffic.json"))[-1]
    committed = json.loads(prof.read_text())
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "profile_secondary.py"), "--check", "affine,upscale,udplanar"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    now = last_json(r.stdout)
    assert len(now) >= 9, now
    off = {k: (v["hbm_bytes_per_launch"], committed[k]["hbm_bytes_per_launch"]) for k, v in now.items()
           if k in committed and abs(v["hbm_bytes_per_launch"] / committed[k]["hbm_bytes_per_launch"] - 1.0) > 0.06}
    assert not off, (prof.name, off)


def test_bench_two_ranks_share_one_gpu(gpu):
    env = dict(os.environ, VALI_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", str(ROOT / "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "16"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 32 and j["scaling"] == "weak"
    assert "cpu_baseline" not in j and j["value"] > 0
    assert j["verification"]["frames_verified_on_gpu"] == 32 and j["verification"]["frames_mismatching"] == 0


def test_bench_gpus_flag_spawns_the_ranks_itself(gpu):
    """`python bench.py --gpus 2` with no launcher (the form the driver uses): bench.py starts the two
    ranks itself; on this one-GPU box they share the device and the collectives run over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["VALI_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--frames", "16"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["launcher"] == "self-spawned"
    assert len(j["per_rank_kernel_ms"]) == 2 and all(v > 0 for v in j["per_rank_kernel_ms"])
    assert j["config"]["global_batch"] == 32 and j["verification"]["frames_verified_on_gpu"] == 32
    assert j["verification"]["frames_mismatching"] == 0 and "cpu_baseline" not in j


def test_bench_eight_ranks_functional_on_one_gpu(gpu):
    """BASELINE configs[4]'s shape (8 ranks, contiguous frame shards, one coefficient broadcast, checksum all-reduce)
    without 8 GPUs: `bench.py --gpus 8` starts its eight ranks itself, they share this box's GPU and the collectives run
    over gloo.  Functional proof only -- no scaling curve has been measured on hardware."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["VALI_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--frames", "8", "--width", "1920",
                        "--height", "1080", "--steps", "2", "--warmup", "1", "--ramp-ms", "0"],
                       capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = last_json(r.stdout)
    assert j["n_gpus"] == 8 and j["ranks_seen"] == 8 and j["launcher"] == "self-spawned"
    assert j["config"]["global_batch"] == 64 and j["config"]["frames_per_gpu"] == 8
    assert len(j["per_rank_kernel_ms"]) == 8 and all(v > 0 for v in j["per_rank_kernel_ms"])
    assert j["verification"] == {"frames_verified_on_gpu": 64, "frames_mismatching": 0,
                                 "checksum_of_checksums": j["verification"]["checksum_of_checksums"]}
    assert j["parity_vs_oracle"]["max_abs_diff_lsb"] == 0
    # the shards: contiguous, disjoint, in rank order, covering the global batch exactly
    assert j["rank_shards"] == [[8 * r_, 8 * r_ + 8] for r_ in range(8)]
    assert "cpu_baseline" not in j and "secondary" not in j


def test_bench_line_carries_the_gpu_state(gpu):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "8",
                        "--cpu-seconds", "0", "--no-secondary", "--ingest-seconds", "0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    st = last_json(r.stdout)["host_placement"]["gpu_state"]
    assert "error" not in st, st
    assert st["under_load"]["samples"] >= 1
    # amdgpu exposes at least the shader clock on every box of the pool; the values are MHz, not Hz or kHz
    lo, hi = st["under_load"]["sclk_mhz"]
    assert 100 <= lo <= hi <= 4000, st


def test_bench_gpus_flag_refuses_more_rccl_ranks_than_gpus(gpu, vali):
    if vali.GetNumGpus() >= 2:
        pytest.skip("box has several GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "VALI_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--frames", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "nothing was launched" in r.stderr        # the parent's pre-flight: no rank was spawned


def test_bench_rccl_path_single_rank(gpu):
    """backend "nccl" (= RCCL) with one rank: process-group init with a device id, the coefficient
    broadcast, the MAX / SUM all-reduces and the barriers all run on the GPU."""
    env = dict(os.environ, VALI_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29618",
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--frames", "16", "--cpu-seconds", "0", "--no-secondary"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = last_json(r.stdout)
    assert j["n_gpus"] == 1 and j["verification"]["frames_mismatching"] == 0
    assert j["parity_vs_oracle"]["max_abs_diff_lsb"] == 0
