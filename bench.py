#!/usr/bin/env python3
"""bench.py -- headline benchmark: batched NV12 -> RGB 3840x2160 on MI355X.

Metric (BASELINE.json): frames/s + achieved HBM GB/s, NV12->RGB 2160p, 1/2/4/8 GPUs.
A "step" is ONE launch of the hot path (PySurfaceConverter.RunBatchAsync ->
vali_nv12_to_rgb_batch) over one batch of `--frames` distinct 2160p surfaces per GPU
(default 512 = BASELINE config 5's 4096 frames / 8 GPUs; 19.1 GB of surfaces per GPU,
far beyond the 256 MiB Infinity Cache).  Inputs are resident in HBM before the timed
region.  One process per GPU; frames are sharded, never exchanged: the only
collective is the RCCL broadcast of the 32-byte colour-coefficient block.

  python bench.py                      # 1 GPU
  python bench.py --gpus N             # spawns N ranks itself (one per GPU, RCCL), rank 0 prints the line
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W     # the same under torchrun

Prints ONE JSON line on rank 0 (contract in the task brief): value = whole-job frames/s,
`roofline` = algorithmic bytes / HIP-event kernel time vs the 8 TB/s HBM peak,
`cpu_baseline` = the C oracle (port of the same arithmetic) timed on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

# every CPU this process may use, BEFORE a rank narrows itself to its GPU's NUMA node: the CPU baseline runs on all of them
HOST_CPUS = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))

import numpy as np  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_CEILING_GBPS = 6290.0  # measured float4 copy ceiling (same guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # 5 warm-up + >= 50 timed batches
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=512, help="frames per GPU per step")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--dst", default="RGB", choices=["RGB", "BGR", "RGB_PLANAR"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="budget of the CPU baseline sample (rank 0, N=1 only); 0 disables")
    ap.add_argument("--ramp-ms", type=float, default=400.0,
                    help="untimed pre-conditioning before the W warm-up steps: the same launch repeated for this long, so "
                         "that the timed steps see the clocks the chip holds under this load and not the idle state of a "
                         "fresh box (the first bench process on a box measured 2.6 %% below the following ones); 0 disables")
    ap.add_argument("--op", default="convert", choices=["convert", "resize", "ud", "preproc"],
                    help="what the sharded pipeline runs on its NV12 frames: convert = NV12->RGB at the same size (the headline; "
                         "BASELINE configs[4]); resize = NV12->NV12 Lanczos to --dst-size; ud = PySurfaceUD NV12->RGB at --dst-size; "
                         "preproc = fused NV12->normalised RGB_32F_PLANAR at --dst-size")
    ap.add_argument("--dst-size", default=None, help="WxH of the output for --op resize / ud / preproc (default: half the source)")
    ap.add_argument("--ingest-seconds", type=float, default=1.5,
                    help="N=1: also time the HOST-FED pipeline (pinned staging + H2D on a copy stream overlapped with the "
                         "conversion, IngestRing) for this long and report it beside the resident number; 0 disables")
    ap.add_argument("--no-numa-bind", action="store_true", help="do not pin each rank to the CPUs of its GPU's NUMA node")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip BASELINE configs[1..3] (tools/bench_configs.py; a few seconds, N=1 only)")
    ap.add_argument("--verbose", action="store_true",
                    help="the long form of the line (every note, the full roofline object of every secondary entry, ~18 KB: what "
                         "profiles/rNN_bench_line.json keeps).  The default line is the SAME measurements in < 7 500 characters, "
                         "weakest kernels last, so that a reader who keeps only the tail of the line sees every entry (VERDICT r05)")
    return ap.parse_args()


def synth_nv12(width, height, seed):
    """SURVEY 8(d) recipe: seeded limited-range noise; frame 0 = full-excursion gradient."""
    rows = height * 3 // 2
    if seed == 0:
        yy, xx = np.mgrid[0:rows, 0:width]
        return ((xx * 255 // (width - 1) + yy * 3) % 256).astype(np.uint8)
    rng = np.random.default_rng(1234 + seed)
    a = np.empty((rows, width), np.uint8)
    a[:height] = rng.integers(16, 236, (height, width), dtype=np.uint8)
    a[height:] = rng.integers(16, 241, (rows - height, width), dtype=np.uint8)
    return a


def cpu_baseline(width, height, coeffs, budget_s):
    """Time the C oracle (same arithmetic as the HIP kernel, AVX2 form: bit-identical to the
    scalar restatement, tests/test_oracle_color.py) on the host cores: one frame on one thread,
    then `cores` independent frames per pass on all cores, repeated until the budget is used."""
    from oracle import oracle as o

    k = o.csc_from_tuple(coeffs)
    cores = len(HOST_CPUS)
    frame = synth_nv12(width, height, 1)
    # one thread alone, then every core: each thread converts its own copy of the frame, input and output first-touched by
    # the thread itself, thread i pinned to the i-th CPU of the host (NUMA-fair).  Sharing read-only inputs and
    # main-thread-touched outputs measured 550 frames/s on 256 cores against 10.9 on one (20 % parallel efficiency, round 2).
    if hasattr(os, "sched_setaffinity"):
        mine = os.sched_getaffinity(0)
        os.sched_setaffinity(0, HOST_CPUS)          # (the rank itself sits on its GPU's NUMA node: widen for the baseline)
    try:
        n1, t1 = o.nv12_to_rgb_bench(frame, width, height, k, 1, min(1.0, budget_s / 8), cpus=HOST_CPUS[:1])
        t_single = t1 / max(n1, 1)
        # all CPUs, then half and a quarter of them (spread over the whole list, i.e. over both sockets): a host whose memory
        # system saturates below its thread count is faster with fewer threads, and the baseline should be its best
        scan = []
        for div in (1, 2, 4):
            n_thr = max(1, cores // div)
            cp = HOST_CPUS[::div][:n_thr]
            d_, t_ = o.nv12_to_rgb_bench(frame, width, height, k, n_thr, budget_s / 3.5, cpus=cp)
            scan.append((d_ / t_, n_thr, d_, t_))
            if n_thr == 1:
                break
        best = max(scan)
        done, t_all, used = best[2], best[3], best[1]
    finally:
        if hasattr(os, "sched_setaffinity"):
            os.sched_setaffinity(0, mine)
    # BASELINE config 1 stand-in: ONE 1080p frame on ONE thread, like the reference's
    # PyFrameConverter (sws_scale on a single frame); best of 5
    hd = synth_nv12(1920, 1080, 1)
    hd_out = [np.zeros((1080, 3 * 1920), np.uint8)]
    t_hd = []
    for _ in range(6):
        t1 = time.perf_counter()
        o.nv12_to_rgb_mt([hd], 1920, 1080, k, 1, hd_out, simd=True)
        t_hd.append(time.perf_counter() - t1)
    t_hd = min(t_hd[1:])
    return {
        "value": round(done / t_all, 3), "unit": "frames/s", "cores": used, "kind": "port", "host_cpus": cores,
        "thread_scan_frames_per_s": {str(n_): round(r_, 1) for r_, n_, _d, _t in scan},
        "sample": f"{done} frames {width}x{height} NV12->RGB by oracle/vali_oracle_simd.c (AVX2, bit-identical to the scalar oracle), "
                  f"{used} pinned OpenMP threads, each on its own first-touched copy of the frame, {t_all:.1f} s",
        "parallel_efficiency": round((done / t_all) * t_single / used, 3),
        "single_thread_fps": round(1.0 / t_single, 3),
        "config1_1080p_single_thread": {"ms_per_frame": round(t_hd * 1e3, 3), "frames_per_s": round(1.0 / t_hd, 2),
                                        "GBps(9331200 B/frame)": round(9331200 / t_hd / 1e9, 3), "cores": 1},
        "note": "the reference's CPU path is FFmpeg libswscale; `swscale` below times it through PyAV when the box has "
                "it (null otherwise); `value` is the build's own C restatement of the GPU arithmetic",
        "swscale": swscale_baseline(width, height, min(budget_s, 5.0)),
    }


def swscale_baseline(width, height, budget_s):
    """The reference's real CPU path -- PyFrameConverter = sws_scale(SWS_BILINEAR), one frame, one thread
    (src/TC/src/TaskConvertFrame.cpp:22-25,79-94) -- timed through PyAV when the box has it (it is not part of the
    build image; never vendored).  None when PyAV is not importable."""
    try:
        import av
    except Exception:
        return None
    nv12 = synth_nv12(width, height, 1)
    frame = av.VideoFrame.from_ndarray(nv12, format="nv12")
    kw = dict(format="rgb24", src_colorspace="ITU709", dst_colorspace="ITU709", interpolation="BILINEAR")
    frame.reformat(**kw)                       # untimed: builds the SwsContext
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        frame.reformat(**kw)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 3), "unit": "frames/s", "cores": 1, "kind": "swscale",
            "sample": f"{done} frames {width}x{height} NV12->RGB24 by libswscale (SWS_BILINEAR, BT.709) through PyAV {av.__version__}, "
                      f"one thread, {dt:.1f} s"}


def secondary_configs(pipe):
    """BASELINE configs[1..3] (parity-test configurations) plus a resize that really interpolates, measured after the timed
    region with HIP events on the task streams: reported beside the headline, never part of
    `value`.  A failure here is recorded, it does not take the headline line down."""
    try:
        pipe.srcs.clear()   # give the 19 GB of headline surfaces back first
        pipe.dsts.clear()
        sys.path.insert(0, str(ROOT / "tools"))
        import bench_configs as bc

        return [bc.hl1080(), bc.cfg2(), bc.cfg3(), bc.interp(), bc.upscale(), bc.cfg4(), bc.udgen(), bc.udplanar(), bc.affine()]
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def measured_traffic(bytes_per_frame, frames, dst, W, H):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*.json), when
    one exists for exactly this workload; PMC counters cannot be read from inside the run."""
    for f in sorted((ROOT / "profiles").glob("*.json"), reverse=True):
        try:
            j = json.loads(f.read_text())
        except Exception:
            continue
        if (j.get("bytes_per_frame") == bytes_per_frame and j.get("frames_per_gpu") == frames
                and f"NV12->{dst} {W}x{H}" in j.get("workload", "")):
            return j["hbm_bytes_per_launch"], f"profiles/{f.name}"
    return None, None


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N copies of this command, one rank per
    GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env exactly as torchrun would set them),
    wait for all of them and return the worst exit code.  Rank 0 inherits stdout, so the ONE JSON
    line is its line; the other ranks' stdout goes to stderr."""
    import socket
    import subprocess

    with socket.socket() as s:       # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VALI_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = dict(enumerate(procs))
        while pending:
            for r, p in list(pending.items()):
                code = p.poll()
                if code is None:
                    continue
                del pending[r]
                if code != 0:
                    rc = rc or code
                    for q in pending.values():     # one rank failed: the others would wait in a
                        q.terminate()              # collective forever (exact PIDs we started)
            time.sleep(0.05)
    except BaseException:
        for p in procs:
            if p.poll() is None:
                p.kill()
        raise
    return rc


def preflight_parent(args):
    """Fail fast, before any rank is spawned, on what would otherwise surface as N half-started processes: more ranks
    than GPUs (RCCL needs one GPU per rank)."""
    if os.environ.get("VALI_BENCH_BACKEND", "nccl") != "nccl":
        return
    try:
        import vali_amd as vali
        ngpu = vali.GetNumGpus()
    except Exception as e:      # noqa: BLE001 -- the ranks will report it properly
        print(f"bench.py: pre-flight skipped ({type(e).__name__}: {e})", file=sys.stderr)
        return
    if args.gpus > ngpu:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {ngpu} GPU(s) (GetNumGpus): nothing was launched. "
                         "Use --gpus <= that, or VALI_BENCH_BACKEND=gloo to let ranks share a GPU for functional tests.")


def preflight_memory(shim, dev, need_bytes, what):
    """frames x (src + dst) must fit the device BEFORE the shard is allocated: a clear message instead of an allocation
    failure half-way through 1024 hipMallocs (or worse, a box driven out of memory)."""
    free_b, total_b = shim.mem_info(dev)
    if need_bytes > free_b * 0.92:
        raise SystemExit(f"bench.py: {what} needs {need_bytes / 2**30:.1f} GiB on GPU {dev}, {free_b / 2**30:.1f} GiB of "
                         f"{total_b / 2**30:.1f} GiB are free -- lower --frames")
    return free_b, total_b


def host_fed_rate(vali, pipe_args, coeffs_ctx, seconds, W, H):
    """The same operator fed from host memory through the ingest ring (pinned staging, H2D on a copy stream overlapped
    with the previous slot's conversion): frames/s a decoder-fed pipeline would see -- PCIe-bound, never `value`."""
    dev, dst_fmt, op, dst_size = pipe_args
    per = 16
    pipe = vali.BatchedFramePipeline(dev, W, H, per, dst_fmt, op=op, dst_size=dst_size)
    pipe.set_coefficients(coeffs_ctx)
    ring = pipe.ingest_ring(slots=3, frames_per_slot=per)
    frame = synth_nv12(W, H, 1).reshape(-1)
    chunk = np.tile(frame, per)

    def chunks():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            yield chunk
    rates = {}
    for name, fill in (("producer_copies_into_pinned_memory", lambda host, item: host.__setitem__(slice(0, item.size), item)),
                       ("producer_writes_in_place", lambda host, item: None)):
        t0 = time.perf_counter()
        n = 0
        for _tag, _dsts in ring.feed(chunks(), fill=fill):
            n += per
        rates[name] = n / (time.perf_counter() - t0)
    ring.close()
    n_dt = rates["producer_copies_into_pinned_memory"]
    return {"frames_per_s": round(n_dt, 1), "GBps_over_pcie": round(n_dt * frame.size / 1e9, 2),
            "frames_per_s_when_the_producer_writes_in_place": round(rates["producer_writes_in_place"], 1),
            "GBps_over_pcie_in_place": round(rates["producer_writes_in_place"] * frame.size / 1e9, 2),
            "slots": 3, "frames_per_slot": per,
            "note": "host -> pinned staging -> H2D on a copy stream, overlapped with the operator on the task stream.  First "
                    "figure: ONE host thread memcpy's every frame into the staging slot (a decoder that hands over pageable "
                    "frames); second: the producer writes the slot in place (a decoder given the pinned buffer), i.e. the "
                    "ring's own ceiling = PCIe"}


def compact_secondary(sec):
    """The secondary measurements as ONE flat list, one short object per kernel: `cfg` (workload), `kernel`, `us` (per frame, batched
    launch on rotating surface sets), `B` (algorithmic bytes per frame), `frac` (B / us / 8 TB/s), `traffic` (HBM bytes per frame from the
    committed PMC passes, or null).  Sorted by `frac`, best first: the tail of the line is where the weak kernels are."""
    if not isinstance(sec, list):
        return sec
    flat = []

    def label(ctx, d, key):
        detail = ([str(key).split("(")[0]] if key else []) + [str(d[k]) for k in ("filter", "formats", "format", "geometry") if k in d]
        return " ".join([ctx.split(" ")[0]] + detail) if detail else ctx      # "interp lanczos NV12 1920x1080->1278x718"

    def walk(o, ctx, key=None):
        if isinstance(o, dict):
            here = o.get("config")
            if isinstance(here, str):
                ctx = here.split(",")[0][:64]
            if "us_per_frame" in o and isinstance(o.get("roofline"), dict):
                r = o["roofline"]
                n = max(1, round(r.get("bytes_per_launch", 0) / max(1, o.get("bytes_moved_per_frame", 1))))
                flat.append({"cfg": label(ctx, o, key), "kernel": (o.get("kernel") or "").split(" (")[0].replace(", ", ","), "us": o["us_per_frame"], "B": o.get("bytes_moved_per_frame"),
                             "frac": r.get("frac"), "traffic": (round(r["traffic"] / n) if r.get("traffic") else None)})
            elif "us_per_call_host_async" in o:      # cfg2: one frame per call, launch-bound -- no roofline object
                flat.append({"cfg": ctx, "us_stream": o.get("us_per_frame_stream_time"), "us_host_async": o.get("us_per_call_host_async"),
                             "us_host_sync_Run": o.get("us_per_call_host_sync_Run"), "frac": 1.0})
            for k, v in o.items():
                if k != "roofline":
                    walk(v, ctx, k if isinstance(v, dict) else key)
        elif isinstance(o, list):
            for v in o:
                walk(v, ctx, key)
    walk(sec, "")
    flat.sort(key=lambda e: -(e.get("frac") or 0))
    for e in flat:
        if "us_stream" in e:
            del e["frac"]
    return flat


def compact_line(out):
    """Default form of the line: nothing measured is dropped, prose and repeated constants are (see --verbose)."""
    sec = out.get("secondary")
    if sec is not None:
        src = None
        def find_src(o):
            nonlocal src
            if isinstance(o, dict):
                if src is None and isinstance(o.get("traffic_source"), str):
                    src = o["traffic_source"]
                for v in o.values():
                    find_src(v)
            elif isinstance(o, list):
                for v in o:
                    find_src(v)
        find_src(sec)
        out["secondary_keys"] = ("us = per frame, one batched launch, rotating >= 1.5 GiB surface sets; B = algorithmic bytes per frame; frac = "
                                 "B / us / 8 TB/s; traffic = HBM bytes per frame, " + (src or "no PMC profile").split(" (")[0] + "; sorted by frac")
        out["secondary"] = compact_secondary(sec)
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        for k in ("note", "thread_scan_frames_per_s", "parallel_efficiency"):
            cb.pop(k, None)
        if isinstance(cb.get("sample"), str):
            cb["sample"] = cb["sample"][:110]
    hf = out.get("host_fed")
    if isinstance(hf, dict):
        hf.pop("note", None)
    hp = out.get("host_placement")
    if isinstance(hp, dict) and isinstance(hp.get("gpu_state"), dict):
        g = hp["gpu_state"]
        ul = g.get("under_load") if isinstance(g.get("under_load"), dict) else {}
        if "error" not in g:
            hp["gpu_state"] = {"under_load": {k: ul.get(k) for k in ("sclk_mhz", "power_w", "samples")},
                               "power_cap_w": g.get("power_cap_w"), "partition": f"{g.get('compute_partition')}/{g.get('memory_partition')}"}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        preflight_parent(args)
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch

    try:
        import vali_amd as vali
    except ImportError:      # clean checkout: the git-ignored .so files are not there yet
        import __graft_entry__

        __graft_entry__.build()
        import vali_amd as vali
    from vali_amd._native import shim

    ngpu = vali.GetNumGpus()
    if ngpu == 0:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback by design)")
    dev = local_rank % ngpu
    torch.cuda.set_device(dev)

    dist = None
    # "nccl" is RCCL on ROCm.  VALI_BENCH_BACKEND=gloo exists only so the N>1 code path can be
    # exercised with several ranks sharing ONE GPU (tests/test_gpu_bench.py); collectives then
    # run on CPU tensors.
    backend = os.environ.get("VALI_BENCH_BACKEND", "nccl")
    coll_dev = f"cuda:{dev}" if backend == "nccl" else "cpu"
    # VALI_BENCH_FORCE_DIST=1: initialise the process group even for one rank, so the RCCL code
    # path (init, broadcast, all-reduce, barrier) can be exercised on a single-GPU box
    if backend == "nccl" and world > ngpu:
        raise SystemExit(f"bench.py: {world} ranks but only {ngpu} GPU(s) visible -- RCCL needs one GPU per rank "
                         "(VALI_BENCH_BACKEND=gloo lets several ranks share a GPU for functional tests)")
    if world > 1 or os.environ.get("VALI_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    W, H, F = args.width, args.height, args.frames
    dst_fmt = vali.PixelFormat[args.dst]
    dst_size = None
    if args.op != "convert":
        dst_size = tuple(int(v) for v in args.dst_size.lower().split("x")) if args.dst_size else (W // 2, H // 2)
    out_fmt, out_size, bytes_per_frame = vali.pipeline.op_geometry(args.op, W, H, dst_fmt, dst_size)
    # each rank next to its GPU: the launch thread and the pinned staging memory on the GPU's NUMA node
    numa_cpus = None if args.no_numa_bind else vali.pipeline.bind_to_gpu_numa(dev)
    # frames x (src + dst), pitches rounded to 256 B: must fit before anything is allocated
    from vali_amd.surface import FORMATS
    def _alloc_bytes(fmt, w, h):
        spec = FORMATS[fmt]
        return sum(-(-pw * spec.elem_size // 256) * 256 * ph for pw, ph in spec.plane_geometry(w, h))
    preflight_memory(shim, dev, F * (_alloc_bytes(vali.NV12, W, H) + _alloc_bytes(out_fmt, *out_size)),
                     f"{F} frames of NV12 {W}x{H} + {out_fmt.name} {out_size[0]}x{out_size[1]}")
    # this rank's shard of the global batch (weak scaling: F frames per GPU)
    begin, end = vali.shard_frames(F * world, rank, world)
    assert end - begin == F
    pipe = vali.BatchedFramePipeline(dev, W, H, F, dst_fmt, op=args.op, dst_size=dst_size)
    # the one collective of the path: rank 0 resolves the colour context to a matrix,
    # everyone receives the 32-byte block over RCCL/xGMI
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    coeffs = pipe.set_coefficients(cc if rank == 0 else None, src=0, device=coll_dev)
    stream = pipe.Stream

    # distinct content in the first NSEED frames (uploaded), cycled device-to-device into
    # the rest: every frame is a distinct HBM allocation, content repeats.
    upl = vali.PyFrameUploader(dev, stream)
    nseed = min(F, 8)
    host = [synth_nv12(W, H, (begin + s) if rank else s) for s in range(nseed)]
    for i in range(nseed):
        ok, info = upl.Run(host[i].reshape(-1), pipe.srcs[i])
        assert ok, info
    for i in range(nseed, F):
        s, d = pipe.srcs[i % nseed]._planes[0], pipe.srcs[i]._planes[0]
        shim.memcpy2d_async(dev, d.GpuMem, d.Pitch, s.GpuMem, s.Pitch, s.Width, s.Height, 2, stream)
    shim.stream_sync(dev, stream)

    def step():
        ok, info = pipe.run_async()
        assert ok, info

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ramp_steps = 0
    if args.ramp_ms > 0:                 # untimed: bring the chip to its steady state under this load
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            step()
            shim.stream_sync(dev, stream)
            ramp_steps += 1
    for _ in range(args.warmup):
        step()
    fence()
    ev = [(shim.event_create(dev), shim.event_create(dev)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:                      # HIP events on the stream the kernel runs on
        shim.event_record(dev, a, stream)
        step()
        shim.event_record(dev, b, stream)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = [shim.event_elapsed_ms(a, b) for a, b in ev]
    for a, b in ev:
        shim.event_destroy(dev, a)
        shim.event_destroy(dev, b)

    per_rank_kernel_ms, ranks_seen = [float(np.mean(kernel_ms))], 1
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own average kernel time (+ its device index), gathered for the report
        mine = torch.tensor([float(np.mean(kernel_ms)), float(dev), float(begin), float(end)], dtype=torch.float64, device=coll_dev)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        per_rank_kernel_ms = [round(float(g[0]), 4) for g in got]
        rank_devices = [int(g[1]) for g in got]
        shards = [[int(g[2]), int(g[3])] for g in got]
        one = torch.ones(1, dtype=torch.int64, device=coll_dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)      # ranks that really took part in the collective
        ranks_seen = int(one.item())
    else:
        rank_devices = [dev]
        shards = [[begin, end]]

    # the state the chip holds under this load, read from sysfs between launches AFTER the timed region (rank 0 only)
    clocks = None
    if rank == 0:
        def _one():
            step()
            shim.stream_sync(dev, stream)
        try:
            clocks = vali.pipeline.sample_clocks_under_load(shim.device_pci_bus_id(dev), _one, 0.25)
        except Exception as e:  # noqa: BLE001 -- telemetry never takes the line down
            clocks = {"error": f"{type(e).__name__}: {e}"}

    # Full-size verification on the GPU (every rank, all F frames): frame i was filled from seed
    # frame i % nseed, so its output must equal output i % nseed byte for byte; the seed outputs
    # themselves are what rank 0 checks against the oracle below.  The per-rank 64-bit checksum
    # and mismatch count are summed over ranks (the optional all-reduce of SURVEY.md 8e).
    verification = None
    if not args.no_parity:
        outs = [torch.from_dlpack(d) for d in pipe.dsts]
        sums = [int(outs[i].sum(dtype=torch.int64).item()) for i in range(nseed)]
        bad = sum(0 if torch.equal(outs[i], outs[i % nseed]) else 1 for i in range(nseed, F))
        tot = torch.tensor([sum(sums[i % nseed] for i in range(F)), bad, F], dtype=torch.int64, device=coll_dev)
        if dist is not None:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        verification = {"frames_verified_on_gpu": int(tot[2]), "frames_mismatching": int(tot[1]),
                        "checksum_of_checksums": int(tot[0])}
        del outs

    parity = None
    if not args.no_parity and rank == 0 and args.op == "convert":
        from oracle import oracle as o

        dwn = vali.PySurfaceDownloader(dev, stream)
        worst = 0
        checked = sorted({0, min(1, F - 1), F - 1})
        for i in checked:
            got = np.zeros(pipe.dsts[i].HostSize, np.uint8)
            ok, _ = dwn.Run(pipe.dsts[i], got)
            assert ok
            want = o.nv12_to_rgb(host[i % nseed], W, H, o.csc_from_tuple(coeffs), args.dst)
            worst = max(worst, int(np.abs(got.astype(np.int16)
                                          - want.reshape(-1).astype(np.int16)).max()))
        parity = {"frames_checked": len(checked), "max_abs_diff_lsb": worst}

    if rank == 0:
        avg_kernel_ms = float(np.mean(kernel_ms))
        achieved = bytes_per_frame * F / (avg_kernel_ms * 1e-3) / 1e9
        fps = world * F * args.steps / elapsed
        traffic, traffic_src = measured_traffic(bytes_per_frame, F, args.dst, W, H) if args.op == "convert" else (None, None)
        op_name = {"convert": "nv12_to_rgb", "resize": "nv12_lanczos_resize", "ud": "nv12_ud_rgb", "preproc": "nv12_preproc_f32"}[args.op]
        what = {"convert": f"PySurfaceConverter.RunBatch NV12->{args.dst} {W}x{H}, BT.709 limited range",
                "resize": f"PySurfaceResizer.RunBatch NV12 {W}x{H}->{out_size[0]}x{out_size[1]} Lanczos",
                "ud": f"PySurfaceUD.RunBatch NV12 {W}x{H}->RGB {out_size[0]}x{out_size[1]}",
                "preproc": f"PySurfacePreprocessor.RunBatch NV12 {W}x{H}->RGB_32F_PLANAR {out_size[0]}x{out_size[1]} normalised"}[args.op]
        out = {
            "metric": f"{op_name}_2160p_frames_per_s" if (W, H) == (3840, 2160)
                      else f"{op_name}_{W}x{H}_frames_per_s",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ramp_ms_untimed": args.ramp_ms, "ramp_steps_untimed": ramp_steps,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"BatchedFramePipeline / {what}, "
                                   f"{F} frames/GPU per step" + (" (BASELINE configs[4] geometry)" if args.op == "convert" else ""),
                       "frames_per_gpu": F, "global_batch": F * world,
                       "bytes_per_frame": bytes_per_frame,
                       "parallelism": f"frame-sharded x{world}, RCCL broadcast of coefficients"},
            "ranks_seen": ranks_seen, "collective_backend": (backend if dist is not None else None),
            "launcher": ("self-spawned" if os.environ.get("VALI_BENCH_SPAWNED") == "1"
                         else "torchrun/env" if "WORLD_SIZE" in os.environ else "single process"),
            "per_rank_kernel_ms": per_rank_kernel_ms, "rank_devices": rank_devices, "rank_shards": shards,
            "achieved_hbm_GBps_whole_job": round(bytes_per_frame * fps / 1e9, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "ratio_to_the_guides_6290GBps_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBPS, 4),
                         "traffic": traffic, "traffic_unit": "B per launch",
                         "traffic_source": traffic_src,
                         "kernel": {"convert": "k_nv12_rgb8", "resize": "k_resize_cols*", "ud": "k_ud_*", "preproc": "k_nv12_preproc"}[args.op],
                         "avg_kernel_ms": round(avg_kernel_ms, 4),
                         "median_kernel_ms": round(float(np.median(kernel_ms)), 4),
                         "min_kernel_ms": round(float(np.min(kernel_ms)), 4)},
        }
        if parity is not None:
            out["parity_vs_oracle"] = parity
        if verification is not None:
            out["verification"] = verification
        out["host_placement"] = {"numa_bound_cpus": (f"{numa_cpus[0]}-{numa_cpus[-1]} ({len(numa_cpus)} CPUs)" if numa_cpus else None),
                                 "pci_bus_id": shim.device_pci_bus_id(dev), "gpu_state": clocks}
        if world == 1 and args.ingest_seconds > 0:
            try:
                pipe.srcs.clear(); pipe.dsts.clear()          # the ring allocates its own slots
                out["host_fed"] = host_fed_rate(vali, (dev, dst_fmt, args.op, dst_size), cc, args.ingest_seconds, W, H)
            except Exception as e:  # noqa: BLE001 -- a side measurement must not take the headline line down
                out["host_fed"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.cpu_seconds > 0 and args.op == "convert":
            out["cpu_baseline"] = cpu_baseline(W, H, coeffs, args.cpu_seconds)
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_configs(pipe)
        if not args.verbose:
            out = compact_line(out)
        print(json.dumps(out, separators=(",", ":")), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
