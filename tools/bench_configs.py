#!/usr/bin/env python3
"""Secondary measurements: BASELINE.json configs[1..3] (the headline configs[4] is bench.py).

  cfg2  PySurfaceConverter NV12->RGB_PLANAR 1920x1080, batch=1 (latency; surface is L3-resident)
  cfg3  PySurfaceResizer 3840x2160 -> 1280x720 bilinear NV12, batch=64 (one launch)
  cfg4  PySurfaceUD NV12 2160p -> RGB 1080p, then PySurfaceRotator 90 degrees (chain)
Prints one JSON object per config.  GB/s figures use the judge's algorithmic bytes of
SURVEY.md 8(d) / BASELINE.md section 3.  Kernel time = HIP events on the task's stream.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vali_amd as vali  # noqa: E402
from vali_amd._native import shim  # noqa: E402

DEV = 0
PEAK = 8000.0


import os

WORKING_SET = 1.5 * 2 ** 30   # distinct surface pairs cycled per timed loop, against the 256 MiB Infinity Cache
# VALI_BENCH_QUICK=1: one small set, two timed launches per entry -- for tools/profile_secondary.py --names (which kernel does
# every entry dispatch?), never for numbers
QUICK = os.environ.get("VALI_BENCH_QUICK", "") not in ("", "0")


def sets_needed(bytes_per_batch):
    """how many distinct src/dst batches a timed loop must rotate through so that it touches >= 1.5 GiB between two
    visits of the same surface (FETCH_SIZE / WRITE_SIZE count Infinity-Cache hits, so the counters cannot show it)"""
    return 1 if QUICK else max(1, -(-int(WORKING_SET) // int(bytes_per_batch)))


def timed(stream, fn, reps, warm=3, warm_s=0.15):
    """HIP-event time per call on `stream` (ms) and host time per call (ms).  `fn` is a callable or a LIST of callables
    that is cycled (one per rotating set of surfaces).  Untimed warm-up: at least `warm` calls AND `warm_s` seconds -- the
    first launches on freshly allocated surfaces of a fresh process measured 8-10 % slow (rotate 2.62 vs 2.40 us, fused UD
    4.19 vs 3.80: clocks and TLBs), whatever the kernel variant."""
    if QUICK:
        reps, warm, warm_s = min(reps, 2), 1, 0.0
    if isinstance(fn, (list, tuple)):
        fns, state = list(fn), [0]
        reps = -(-reps // len(fns)) * len(fns)    # whole cycles

        def fn():
            fns[state[0] % len(fns)]()
            state[0] += 1
        warm = max(warm, len(fns))
    t_w = time.perf_counter()
    k = 0
    while k < warm or time.perf_counter() - t_w < warm_s:
        fn()
        k += 1
        if k % 8 == 0:
            shim.stream_sync(DEV, stream)
    shim.stream_sync(DEV, stream)
    a, b = shim.event_create(DEV), shim.event_create(DEV)
    t0 = time.perf_counter()
    shim.event_record(DEV, a, stream)
    for _ in range(reps):
        fn()
    shim.event_record(DEV, b, stream)
    shim.event_sync(DEV, b)
    wall = (time.perf_counter() - t0) / reps
    ms = shim.event_elapsed_ms(a, b) / reps
    shim.event_destroy(DEV, a)
    shim.event_destroy(DEV, b)
    return ms, wall * 1e3


def fill(surfs, seed=0):
    rng = np.random.default_rng(seed)
    up = vali.PyFrameUploader(DEV)
    host = rng.integers(16, 236, surfs[0].HostSize, dtype=np.uint8)
    for i, s in enumerate(surfs):
        if i < 4:
            assert up.Run(np.roll(host, i * 977), s)[0]
        else:
            p, q = surfs[i % 4]._planes, s._planes
            for a, b in zip(p, q):
                shim.memcpy2d_async(DEV, b.GpuMem, b.Pitch, a.GpuMem, a.Pitch, a.Width * a.ElemSize, a.Height, 2, 0)
    shim.stream_sync(DEV, 0)


def hl1080(n=1024):
    """the headline conversion on 1080p surfaces (north_star asks for 1080p next to 2160p): NV12 -> RGB,
    one launch over n frames resident in HBM"""
    w, h = 1920, 1080
    srcs = [vali.Surface.Make(vali.NV12, w, h, DEV) for _ in range(n)]
    dsts = [vali.Surface.Make(vali.RGB, w, h, DEV) for _ in range(n)]
    fill(srcs)
    cvt = vali.PySurfaceConverter(DEV)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    batch = cvt.PrepareBatch(srcs, dsts)
    ms, _ = timed(cvt.Stream, lambda: cvt.RunBatchAsync(batch, cc_ctx=cc), 20)
    return {"config": f"NV12->RGB 1920x1080, batch={n}, one launch (the headline kernel at 1080p)", "kernel": "k_nv12_rgb8",
            "frames_per_s": round(n / (ms * 1e-3), 1), "us_per_frame": round(ms * 1e3 / n, 3),
            "bytes_moved_per_frame": int(w * h * 4.5), "roofline": roofline("hl1080", w * h * 4.5, n, ms, 1)}


def cfg2():
    w, h = 1920, 1080
    cvt = vali.PySurfaceConverter(DEV)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    src = [vali.Surface.Make(vali.NV12, w, h, DEV)]
    fill(src)
    dst = vali.Surface.Make(vali.RGB_PLANAR, w, h, DEV)
    mid = vali.Surface.Make(vali.RGB, w, h, DEV)
    ms_async, wall_async = timed(cvt.Stream, lambda: cvt.RunAsync(src[0], dst, cc), 2000, 50)
    t0 = time.perf_counter()
    for _ in range(500):
        cvt.Run(src[0], dst, cc)
    wall_sync = (time.perf_counter() - t0) / 500 * 1e3
    ms_chain, _ = timed(cvt.Stream, lambda: (cvt.RunAsync(src[0], mid, cc), cvt.RunAsync(mid, dst)), 1000, 50)
    # the same per-frame chains replayed from a hipGraph (StreamCapture): one submission per frame
    small = vali.Surface.Make(vali.NV12, 960, 540, DEV)
    rgb_s = vali.Surface.Make(vali.RGB, 960, 540, DEV)
    f32_s = vali.Surface.Make(vali.RGB_32F, 960, 540, DEV)
    pl_s = vali.Surface.Make(vali.RGB_32F_PLANAR, 960, 540, DEV)
    rs = vali.PySurfaceResizer(vali.NV12, DEV, cvt.Stream, interpolation=vali.Interpolation.LINEAR)

    def chain4():
        rs.RunAsync(src[0], small); cvt.RunAsync(small, rgb_s, cc); cvt.RunAsync(rgb_s, f32_s); cvt.RunAsync(f32_s, pl_s)
    chain4()
    ms_c4, wall_c4 = timed(cvt.Stream, chain4, 1000, 50)
    cap2 = vali.StreamCapture(cvt.Stream, DEV)
    with cap2:
        cvt.RunAsync(src[0], mid, cc); cvt.RunAsync(mid, dst)
    ms_g2, wall_g2 = timed(cvt.Stream, cap2.Launch, 1000, 50)
    cap4 = vali.StreamCapture(cvt.Stream, DEV)
    with cap4:
        chain4()
    ms_g4, wall_g4 = timed(cvt.Stream, cap4.Launch, 1000, 50)
    graph = {"chain2_eager_us": round(ms_chain * 1e3, 3), "chain2_graph_us": round(ms_g2 * 1e3, 3),
             "chain4(resize+cvt+f32+planar)_eager_us": round(ms_c4 * 1e3, 3), "chain4_graph_us": round(ms_g4 * 1e3, 3),
             "chain4_eager_host_us": round(wall_c4 * 1e3, 3), "chain4_graph_host_us": round(wall_g4 * 1e3, 3)}
    b = 9331200
    return {"config": "cfg2 PySurfaceConverter NV12->RGB_PLANAR 1920x1080 batch=1 (fused single kernel)",
            "hipgraph_replay": graph,
            "us_per_frame_stream_time": round(ms_async * 1e3, 3), "us_per_call_host_async": round(wall_async * 1e3, 3),
            "us_per_call_host_sync_Run": round(wall_sync * 1e3, 3),
            "GBps_algorithmic": round(b / (ms_async * 1e-3) / 1e9, 1),
            "note": "batch=1 working set (9.3 MB) lives in the 256 MiB Infinity Cache; time is launch latency, not HBM",
            "reference_faithful_2step_chain_us": round(ms_chain * 1e3, 3)}


TRAFFIC = None


def roofline(key, bytes_per_frame, n, ms, sets=1):
    """HBM roofline entry of one secondary kernel: `achieved` = the bytes the kernel really has to move (stated per
    config) / its HIP-event time; `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes
    (profiles/r02_secondary_traffic.json, tools/profile_secondary.py), when there is an entry for this config."""
    global TRAFFIC
    if TRAFFIC is None:
        prof = Path(__file__).resolve().parent.parent / "profiles"
        f = next(iter(sorted(prof.glob("r*_secondary_traffic.json"), reverse=True)), None)     # the newest round's passes
        TRAFFIC = json.loads(f.read_text()) if f else {}
    gbps = bytes_per_frame * n / (ms * 1e-3) / 1e9
    t = TRAFFIC.get(key, {})
    return {"bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK, "unit": "GB/s", "frac": round(gbps / PEAK, 4),
            "bytes_per_launch": int(bytes_per_frame * n),
            "working_set_bytes": int(bytes_per_frame * n * sets), "surface_sets_cycled": sets,
            "traffic": t.get("hbm_bytes_per_launch") if t.get("frames") == n else None,
            "traffic_source": t.get("source") if t.get("frames") == n else None}


def make_sets(k, make):
    """k independent sets of whatever `make()` builds (surfaces + a prepared batch)"""
    return [make() for _ in range(k)]


def cfg3(n=64):
    sw, sh, dw, dh = 3840, 2160, 1280, 720
    rs = vali.PySurfaceResizer(vali.NV12, DEV, interpolation=vali.Interpolation.LINEAR)   # config 3 names bilinear
    touched = 720 * 3840 + 360 * 3840 + 1382400  # one source row per dst row (weight of the 2nd is 0) + dst
    k = sets_needed(touched * n)

    def make():
        srcs = [vali.Surface.Make(vali.NV12, sw, sh, DEV) for _ in range(n)]
        dsts = [vali.Surface.Make(vali.NV12, dw, dh, DEV) for _ in range(n)]
        fill(srcs)
        return srcs, dsts, rs.PrepareBatch(srcs, dsts)
    sets = make_sets(k, make)
    ms, wall = timed(rs.Stream, [lambda b=b: rs.RunBatchAsync(b) for _, _, b in sets], 50)
    return {"config": f"cfg3 PySurfaceResizer NV12 3840x2160->1280x720 bilinear, batch={n}, one launch",
            "kernel": "k_resize_pointk<3>", "ms_per_batch": round(ms, 4), "us_per_frame": round(ms * 1e3 / n, 3),
            "frames_per_s": round(n / (ms * 1e-3), 1),
            "bytes_moved_per_frame": touched,
            "bytes_note": "exact 3x: the NPP grid samples src[3y][3x] (weights 1,0); the kernel fetches the 1080 source rows it "
                          "samples (4 147 200 B) and writes 1 382 400 B; SURVEY 8d's 13 824 000 B assumes the whole source is read",
            "roofline": roofline("cfg3", touched, n, ms, k),
            "speedup_vs_reading_the_whole_source_at_8TBps": round((13824000 / 8e12) / (ms * 1e-3 / n), 3)}


def interp(n=64):
    """A resize that really interpolates: NV12 2160p -> 1920x1088 (every source row contributes; exactly 2:1 along x, so the
    Lanczos kernel takes its 2:1-along-x form), -> 1936x1088 (no integer ratio on either axis: the general form, on
    specialised waves since round 5), 1080p -> 720p (3:2 both ways: the uniform-weight form) and the general form below
    2160p -- NV12 1080p -> 1278x718 and packed RGB 1080p -> 1277x719 -- with the bilinear filter of BASELINE
    config 3 and the reference's own filter (Lanczos-3, the PySurfaceResizer default)."""
    out = []
    for (fmt, sw, sh, dw, dh) in ((vali.NV12, 3840, 2160, 1920, 1088), (vali.NV12, 3840, 2160, 1936, 1088), (vali.NV12, 1920, 1080, 1280, 720),
                                  (vali.NV12, 1920, 1080, 1278, 718), (vali.RGB, 1920, 1080, 1277, 719)):
        b = (sw * sh + dw * dh) * 3 // 2 if fmt == vali.NV12 else (sw * sh + dw * dh) * 3
        for name, it in (("bilinear", vali.Interpolation.LINEAR), ("lanczos", vali.Interpolation.LANCZOS)):
            if (dw, name) != (1920, "bilinear") and name == "bilinear":
                continue
            rs = vali.PySurfaceResizer(fmt, DEV, interpolation=it)
            k = sets_needed(b * n)

            def make():
                srcs = [vali.Surface.Make(fmt, sw, sh, DEV) for _ in range(n)]
                dsts = [vali.Surface.Make(fmt, dw, dh, DEV) for _ in range(n)]
                fill(srcs)
                return srcs, dsts, rs.PrepareBatch(srcs, dsts)
            sets = make_sets(k, make)
            ms, _ = timed(rs.Stream, [lambda q=q: rs.RunBatchAsync(q) for _, _, q in sets], 20)
            kern = "k_resize<u8, 2>" if name == "bilinear" else {1920: "k_resize_cols_x2<u8, 12, 6, 3>", 1936: "k_resize_cols_ws<u8, 12, 6, 3, 4>",
                                                                  1280: "k_resize_cols_x32<u8, 12, 6, static rows>",
                                                                  1278: "k_resize_cols_ws<u8, 12, 6, 4, 5>", 1277: "k_resize_cols_ws<u8, 3, 6, 4, 5>"}[dw]
            key = "interp_" + name + {1920: "", 1936: "_1936", 1280: "_720p", 1278: "_1278", 1277: "_rgb_1277"}[dw]
            out.append({"filter": name, "format": fmt.name, "geometry": f"{sw}x{sh}->{dw}x{dh}", "kernel": kern,
                        "us_per_frame": round(ms * 1e3 / n, 3), "bytes_moved_per_frame": b,
                        "roofline": roofline(key, b, n, ms, k)})
            del sets
    return {"config": f"interp PySurfaceResizer NV12 3840x2160->1920x1088 / 1936x1088, 1920x1080->1280x720 / 1278x718, RGB 1920x1080->1277x719 (non-integer ratios), batch={n}, one launch per filter",
            "bytes_note": "whole source + destination per frame", "results": out}


def upscale(n=64):
    """Planes that GROW (the upscale to display size) under the reference's filter: NV12 720p -> 1080p (exactly 3:2 both ways:
    the static form k_resize_rows_x23) and 720p -> 1600x900 (5:4, no special form: k_resize_rows_reg, the row pass and the vertical
    window in registers, no LDS stage); packed RGB 720p -> 1080p (round 5: the 3:2 form's three-channel variant, k_resize_rows_x23_rgb)
    and 720p -> 1600x900 (round 5: the register form for three channels, k_resize_rows_rgb); and the two forms the fast paths do not
    cover: RGB_32F 720p -> 1080p (k_resize_taps) and P10 720p -> 1600x900 (k_resize_rows, the LDS-staged rows form)."""
    out = []
    for (fmt, sw, sh, dw, dh) in ((vali.NV12, 1280, 720, 1920, 1080), (vali.NV12, 1280, 720, 1600, 900), (vali.RGB, 1280, 720, 1920, 1080),
                                  (vali.RGB, 1280, 720, 1600, 900), (vali.RGB_32F, 1280, 720, 1920, 1080), (vali.P10, 1280, 720, 1600, 900)):
        rs = vali.PySurfaceResizer(fmt, DEV)                       # Lanczos-3, the reference's (and the task's default) filter
        b = {vali.NV12: (sw * sh + dw * dh) * 3 // 2, vali.RGB: (sw * sh + dw * dh) * 3, vali.P10: (sw * sh + dw * dh) * 3,
             vali.RGB_32F: (sw * sh + dw * dh) * 12}[fmt]
        k = sets_needed(b * n)

        def make():
            srcs = [vali.Surface.Make(fmt, sw, sh, DEV) for _ in range(n)]
            dsts = [vali.Surface.Make(fmt, dw, dh, DEV) for _ in range(n)]
            fill(srcs)
            return srcs, dsts, rs.PrepareBatch(srcs, dsts)
        sets = make_sets(k, make)
        ms, _ = timed(rs.Stream, [lambda q=q: rs.RunBatchAsync(q) for _, _, q in sets], 30)
        kern = ("k_resize_rows_x23_rgb<6, 48>" if fmt == vali.RGB and 3 * sw == 2 * dw else "k_resize_rows_rgb<64>" if fmt == vali.RGB
                else "k_resize_taps<f32, 3, 6>" if fmt == vali.RGB_32F
                else "k_resize_rows<u16, 12, 6, 32>" if fmt == vali.P10
                else "k_resize_rows_x23<u8, 12, 6, 48>" if 3 * sw == 2 * dw else "k_resize_rows_reg<64, 1, 2>")
        key = f"upscale_{dw}x{dh}" + ("" if fmt == vali.NV12 else "_" + fmt.name.lower())
        out.append({"filter": "lanczos", "format": fmt.name, "geometry": f"{sw}x{sh}->{dw}x{dh}", "kernel": kern,
                    "us_per_frame": round(ms * 1e3 / n, 3), "bytes_moved_per_frame": b,
                    "roofline": roofline(key, b, n, ms, k)})
        del sets
    return {"config": f"upscale PySurfaceResizer 1280x720 -> 1920x1080 / 1600x900 Lanczos (planes that grow: NV12, packed RGB, P10), batch={n}, one launch",
            "bytes_note": "whole source + destination per frame", "results": out}


def affine(n=64):
    """PySurfaceRotator at an angle that is no quarter turn (the reference's nppiRotate: RotateSurface.cpp:22-159): RGB 1080p by 30
    degrees, bilinear, destination of the same size (k_rotate_affine_lds: the tile's source box staged in LDS, round 6)."""
    w, h = 1920, 1080
    rot = vali.PySurfaceRotator(DEV)
    b = 2 * w * h * 3
    k = sets_needed(b * n)

    def make():
        srcs = [vali.Surface.Make(vali.RGB, w, h, DEV) for _ in range(n)]
        dsts = [vali.Surface.Make(vali.RGB, w, h, DEV) for _ in range(n)]
        fill(srcs)
        return srcs, dsts, rot.PrepareBatch(srcs, dsts)
    sets = make_sets(k, make)
    ms, _ = timed(rot.Stream, [lambda q=q: rot.RunBatchAsync(q, angle=30.0, shift_x=0.0, shift_y=0.0) for _, _, q in sets], 30)
    # ... and about the CENTRE of the frame (what a user means by "rotate by 30 degrees"): 80 % of the destination samples inside the
    # source instead of 52 % -- the same kernel with more of its tiles at work
    import math
    a, cx, cy = math.radians(30.0), (w - 1) / 2.0, (h - 1) / 2.0
    sx, sy = cx - (cx * math.cos(a) + cy * math.sin(a)), cy - (-cx * math.sin(a) + cy * math.cos(a))
    ms_c, _ = timed(rot.Stream, [lambda q=q: rot.RunBatchAsync(q, angle=30.0, shift_x=sx, shift_y=sy) for _, _, q in sets], 30)
    note = "whole source + destination (destination pixels that sample outside the source are neither fetched nor stored: see traffic)"
    return {"config": f"affine PySurfaceRotator RGB 1920x1080 by 30 degrees (bilinear), batch={n}, one launch", "bytes_note": note,
            "results": [{"geometry": "about the origin (shifts 0, 0)", "kernel": "k_rotate_affine_lds<u8, 64, 4>", "us_per_frame": round(ms * 1e3 / n, 3),
                         "bytes_moved_per_frame": b, "roofline": roofline("affine_rgb_30", b, n, ms, k)},
                        {"geometry": "about the centre", "kernel": "k_rotate_affine_lds<u8, 64, 4>", "us_per_frame": round(ms_c * 1e3 / n, 3),
                         "bytes_moved_per_frame": b, "roofline": roofline("affine_rgb_30_centre", b, n, ms_c, k)}]}


def cfg4(n=64):
    sw, sh, dw, dh = 3840, 2160, 1920, 1080
    ud = vali.PySurfaceUD(DEV)
    rot = vali.PySurfaceRotator(DEV, ud.Stream)
    b_ud, b_rot = 18662400, 12441600
    k = sets_needed(b_rot * n)    # the smallest of the three working sets decides (2 sets: 1.6 / 2.4 GB)

    def make():
        srcs = [vali.Surface.Make(vali.NV12, sw, sh, DEV) for _ in range(n)]
        mids = [vali.Surface.Make(vali.RGB, dw, dh, DEV) for _ in range(n)]
        outs = [vali.Surface.Make(vali.RGB, dh, dw, DEV) for _ in range(n)]
        fill(srcs)
        return srcs, mids, outs, ud.PrepareBatch(srcs, mids), rot.PrepareBatch(mids, outs), ud.PrepareBatch(srcs, outs)
    sets = make_sets(k, make)
    ms_ud, _ = timed(ud.Stream, [lambda q=q[3]: ud.RunBatchAsync(q) for q in sets], 30)
    ms_rot, wall_rot = timed(ud.Stream, [lambda q=q[4]: rot.RunBatchAsync(q, angle=90.0) for q in sets], 30)
    mids, outs = sets[0][1], sets[0][2]
    ms_rot1, wall_rot1 = timed(ud.Stream, lambda: rot.RunAsync(mids[0], outs[0], 90.0), 200, 20)
    ms_fused, _ = timed(ud.Stream, [lambda q=q[5]: ud.RunRotatedBatchAsync(q, angle=90.0) for q in sets], 30)
    return {"config": f"cfg4 PySurfaceUD NV12 2160p->RGB 1080p (batch={n}, one launch) + PySurfaceRotator 90deg (batch, one launch)",
            "ud": {"kernel": "k_ud_half<RGB>", "us_per_frame": round(ms_ud * 1e3 / n, 3), "bytes_moved_per_frame": b_ud,
                   "roofline": roofline("cfg4_ud", b_ud, n, ms_ud, k)},
            "rot": {"kernel": "k_rotate_tile<3, 90>", "us_per_frame": round(ms_rot * 1e3 / n, 3), "bytes_moved_per_frame": b_rot,
                    "roofline": roofline("cfg4_rot", b_rot, n, ms_rot, k),
                    "single_call_us(stream)": round(ms_rot1 * 1e3, 3), "single_call_us(host)": round(wall_rot1 * 1e3, 3)},
            "chain": {"us_per_frame": round((ms_ud + ms_rot) * 1e3 / n, 3), "bytes_moved_per_frame": b_ud + b_rot,
                      "roofline": roofline("cfg4_chain", b_ud + b_rot, n, ms_ud + ms_rot, k)},
            "fused(PySurfaceUD.RunRotatedBatch)": {"kernel": "k_ud_half_t<90, 64>", "us_per_frame": round(ms_fused * 1e3 / n, 3),
                                                   "bytes_moved_per_frame": b_ud,
                                                   "bytes_note": "one pass: the 6 220 800 B intermediate is neither written nor re-read",
                                                   "roofline": roofline("cfg4_fused", b_ud, n, ms_fused, k),
                                                   "speedup_vs_chain": round((ms_ud + ms_rot) / ms_fused, 3)}}


def udgen(n=64):
    """PySurfaceUD at ratios the exact-2x kernel does not cover (the any-ratio kernel k_ud_nv12, staged form): the
    pre-processing geometries of a 1080p stream -- 720p (1.5x), 640x384 (3x / 2.8x) -- and the unchanged size (colour conversion
    with interpolated chroma: k_ud_lean), NV12 -> packed RGB."""
    out = []
    ud = vali.PySurfaceUD(DEV)
    for (sw, sh, dw, dh) in ((1920, 1080, 1280, 720), (1920, 1080, 640, 384), (1920, 1080, 1920, 1080)):
        b = sw * sh * 3 // 2 + dw * dh * 3
        k = sets_needed(b * n)

        def make():
            srcs = [vali.Surface.Make(vali.NV12, sw, sh, DEV) for _ in range(n)]
            dsts = [vali.Surface.Make(vali.RGB, dw, dh, DEV) for _ in range(n)]
            fill(srcs)
            return srcs, dsts, ud.PrepareBatch(srcs, dsts)
        sets = make_sets(k, make)
        ms, _ = timed(ud.Stream, [lambda q=q: ud.RunBatchAsync(q) for _, _, q in sets], 30)
        out.append({"geometry": f"{sw}x{sh}->{dw}x{dh}", "kernel": "k_ud_32<RGB>" if 2 * sw == 3 * dw and 2 * sh == 3 * dh else "k_ud_lean<RGB, 1, even>" if (sw, sh) == (dw, dh) else "k_ud_nv12<u8, RGB, staged>", "us_per_frame": round(ms * 1e3 / n, 3),
                    "bytes_moved_per_frame": b, "roofline": roofline(f"udgen_{dw}x{dh}", b, n, ms, k)})
        del sets
    return {"config": f"udgen PySurfaceUD NV12 1080p -> RGB at non-2x ratios, batch={n}, one launch each",
            "bytes_note": "whole NV12 source + RGB destination", "results": out}


def udplanar(n=64):
    """PySurfaceUD on planar sources (the reference's UDPlanar, UDSurface.cpp:33-93: every plane through NPP Lanczos to the
    size of the matching dst plane) at unchanged size -- YUV420 -> YUV444: luma 1:1 (k_plane_copy), chroma exactly
    doubled (k_resize_up2) -- and the 10-bit pair of the reference's own golden."""
    out = []
    ud = vali.PySurfaceUD(DEV)
    for (sf, df, w, h, eb) in ((vali.YUV420, vali.YUV444, 1920, 1080, 1), (vali.YUV420_10bit, vali.YUV444_10bit, 1920, 1080, 2)):
        b = (w * h * 3 // 2 + w * h * 3) * eb
        k = sets_needed(b * n)

        def make():
            srcs = [vali.Surface.Make(sf, w, h, DEV) for _ in range(n)]
            dsts = [vali.Surface.Make(df, w, h, DEV) for _ in range(n)]
            fill(srcs)
            return srcs, dsts, ud.PrepareBatch(srcs, dsts)
        sets = make_sets(k, make)
        ms, _ = timed(ud.Stream, [lambda q=q: ud.RunBatchAsync(q) for _, _, q in sets], 30)
        out.append({"formats": f"{sf.name}->{df.name}", "geometry": f"{w}x{h}->{w}x{h}", "kernel": "k_resize_up2<T, 6> (U, V; the luma copy rides in the same launch)",
                    "us_per_frame": round(ms * 1e3 / n, 3), "bytes_moved_per_frame": b,
                    "roofline": roofline(f"udplanar_{eb * 8}bit", b, n, ms, k)})
        del sets
    return {"config": f"udplanar PySurfaceUD YUV420 -> YUV444 1080p at unchanged size (Lanczos like UDPlanar), batch={n}, one launch",
            "bytes_note": "whole 4:2:0 source + 4:4:4 destination", "results": out}


def cfg3_lanczos(n=64):
    """cfg3 geometries with the reference's own filter (Lanczos-3) -- exact 3x and a non-integer factor"""
    out = []
    for (sw, sh, dw, dh) in ((3840, 2160, 1280, 720), (3840, 2160, 1920, 1088), (1920, 1080, 3840, 2160)):
        res = {}
        for name, interp in (("linear", vali.Interpolation.LINEAR), ("cubic", vali.Interpolation.CUBIC),
                             ("lanczos", vali.Interpolation.LANCZOS)):
            rs = vali.PySurfaceResizer(vali.NV12, DEV, interpolation=interp)
            m = n if sw * sh <= 3840 * 2160 and dw * dh <= 1920 * 1088 else 16
            srcs = [vali.Surface.Make(vali.NV12, sw, sh, DEV) for _ in range(m)]
            dsts = [vali.Surface.Make(vali.NV12, dw, dh, DEV) for _ in range(m)]
            fill(srcs)
            batch = rs.PrepareBatch(srcs, dsts)
            ms, _ = timed(rs.Stream, lambda: rs.RunBatchAsync(batch), 20)
            b = (sw * sh + dw * dh) * 3 // 2
            res[name] = {"us_per_frame": round(ms * 1e3 / m, 3), "GBps_src_plus_dst": round(b * m / (ms * 1e-3) / 1e9, 1)}
        out.append({"geometry": f"NV12 {sw}x{sh}->{dw}x{dh}", **res})
    return {"config": "resize filters", "results": out}


def preproc(n=64):
    """fused NV12 -> resize -> RGB -> f32 planar normalised vs the surface part of the
    reference-style chain (resizer + 3 converter launches; the two torch kernels of the chain are
    NOT included in the chain time, so the real gap is larger)."""
    out = []
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    for (sw, sh, dw, dh) in ((1920, 1080, 1920, 1080), (3840, 2160, 640, 640), (1920, 1080, 640, 384)):
        srcs = [vali.Surface.Make(vali.NV12, sw, sh, DEV) for _ in range(n)]
        fill(srcs)
        dsts = [vali.Surface.Make(vali.RGB_32F_PLANAR, dw, dh, DEV) for _ in range(n)]
        pp = vali.PySurfacePreprocessor(DEV, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), div=255.0)
        b = pp.PrepareBatch(srcs, dsts)
        ms_f, _ = timed(pp.Stream, lambda: pp.RunBatchAsync(b, cc_ctx=cc), 20)
        # chain
        rs = vali.PySurfaceResizer(vali.NV12, DEV, pp.Stream, interpolation=vali.Interpolation.LINEAR)
        cv = vali.PySurfaceConverter(DEV, pp.Stream)
        small = srcs if (sw, sh) == (dw, dh) else [vali.Surface.Make(vali.NV12, dw, dh, DEV) for _ in range(n)]
        rgb = [vali.Surface.Make(vali.RGB, dw, dh, DEV) for _ in range(n)]
        f32 = [vali.Surface.Make(vali.RGB_32F, dw, dh, DEV) for _ in range(n)]
        b0 = rs.PrepareBatch(srcs, small) if small is not srcs else None
        b1, b2, b3 = cv.PrepareBatch(small, rgb), cv.PrepareBatch(rgb, f32), cv.PrepareBatch(f32, dsts)

        def chain():
            if b0 is not None:
                rs.RunBatchAsync(b0)
            cv.RunBatchAsync(b1, cc_ctx=cc)
            cv.RunBatchAsync(b2)
            cv.RunBatchAsync(b3)
        ms_c, _ = timed(pp.Stream, chain, 20)
        alg = sw * sh * 3 // 2 + dw * dh * 12
        out.append({"geometry": f"NV12 {sw}x{sh} -> RGB_32F_PLANAR {dw}x{dh} normalised, batch {n}",
                    "fused_us_per_frame": round(ms_f * 1e3 / n, 3), "chain_us_per_frame(surface part only)": round(ms_c * 1e3 / n, 3),
                    "speedup": round(ms_c / ms_f, 2), "fused_GBps(src+dst bytes)": round(alg * n / (ms_f * 1e-3) / 1e9, 1)})
    return {"config": "fused pre-processing (PySurfacePreprocessor)", "results": out}


def ud_scales(n=32):
    """UD per destination pixel at different scale factors (LDS bank-conflict / staging study)."""
    out = []
    ud = vali.PySurfaceUD(DEV)
    for (sw, sh, dw, dh) in ((3840, 2160, 1920, 1080), (1920, 1080, 1920, 1080), (5760, 3240, 1920, 1080),
                             (2880, 1620, 1920, 1080), (960, 540, 1920, 1080), (2560, 1440, 1920, 1080)):
        srcs = [vali.Surface.Make(vali.NV12, sw, sh, DEV) for _ in range(n)]
        dsts = [vali.Surface.Make(vali.RGB, dw, dh, DEV) for _ in range(n)]
        fill(srcs)
        b = ud.PrepareBatch(srcs, dsts)
        ms, _ = timed(ud.Stream, lambda: ud.RunBatchAsync(b), 20)
        out.append({"geometry": f"{sw}x{sh}->{dw}x{dh}", "scale": round(sw / dw, 3), "us_per_frame": round(ms * 1e3 / n, 3),
                    "GBps": round((sw * sh * 1.5 + dw * dh * 3) * n / (ms * 1e-3) / 1e9, 1)})
    return {"config": "UD NV12->RGB 1080p output at several source scales", "results": out}


if __name__ == "__main__":
    which = sys.argv[1:] or ["hl1080", "cfg2", "cfg3", "interp", "upscale", "cfg4", "udgen", "udplanar", "affine"]
    for name in which:
        print(json.dumps(globals()[name]()), flush=True)
