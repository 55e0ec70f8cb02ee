#!/usr/bin/env python3
"""Per-call cost of every operator on ONE frame (RunAsync back to back on one stream: stream time and host time per
call, us) next to its per-frame cost inside a batch of 64 -- the per-frame use of the reference's API (BASELINE config 2)
at 1080p and 2160p.  A call cannot be cheaper than the launch path (~4 us); what this shows is which kernels leave the
chip idle when they get a single frame."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill

cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
L, Li = vali.Interpolation.LANCZOS, vali.Interpolation.LINEAR


def run(name, mk, fs, fd, s, d, single, batch):
    t = mk()
    n = 32
    srcs = [vali.Surface.Make(fs, s[0], s[1], DEV) for _ in range(n)]
    dsts = [vali.Surface.Make(fd, d[0], d[1], DEV) for _ in range(n)]
    fill(srcs)
    ms1, wall1 = timed(t.Stream, lambda: single(t, srcs[0], dsts[0]), 200, 20)
    b = t.PrepareBatch(srcs, dsts)
    msb, _ = timed(t.Stream, lambda: batch(t, b), 10, 2, 0.05)
    print(f"{name:34s} {s[0]}x{s[1]}->{d[0]}x{d[1]}: single {ms1 * 1e3:6.2f} us stream / {wall1 * 1e3:5.2f} us host; in a batch {msb * 1e3 / n:6.2f} us", flush=True)


for (w, h) in ((1920, 1080), (3840, 2160)):
    hw, hh = w // 2, (h // 2 + 15) // 16 * 16 if h == 2160 else h // 2
    run("convert NV12->RGB", lambda: vali.PySurfaceConverter(DEV), vali.NV12, vali.RGB, (w, h), (w, h), lambda t, a, b: t.RunAsync(a, b, cc), lambda t, b: t.RunBatchAsync(b, cc))
    run("convert RGB->YUV420", lambda: vali.PySurfaceConverter(DEV), vali.RGB, vali.YUV420, (w, h), (w, h), lambda t, a, b: t.RunAsync(a, b), lambda t, b: t.RunBatchAsync(b))
    run("resize bilinear NV12 (non-integer)", lambda: vali.PySurfaceResizer(vali.NV12, DEV, interpolation=Li), vali.NV12, vali.NV12, (w, h), (hw, hh + 8), lambda t, a, b: t.RunAsync(a, b), lambda t, b: t.RunBatchAsync(b))
    run("resize Lanczos NV12 (non-integer)", lambda: vali.PySurfaceResizer(vali.NV12, DEV, interpolation=L), vali.NV12, vali.NV12, (w, h), (hw, hh + 8), lambda t, a, b: t.RunAsync(a, b), lambda t, b: t.RunBatchAsync(b))
    run("resize bilinear NV12 3x (point)", lambda: vali.PySurfaceResizer(vali.NV12, DEV, interpolation=Li), vali.NV12, vali.NV12, (w, h), (w // 3, h // 3), lambda t, a, b: t.RunAsync(a, b), lambda t, b: t.RunBatchAsync(b))
    run("UD NV12->RGB 2x", lambda: vali.PySurfaceUD(DEV), vali.NV12, vali.RGB, (w, h), (w // 2, h // 2), lambda t, a, b: t.RunAsync(a, b), lambda t, b: t.RunBatchAsync(b))
    run("UD NV12->RGB 1.5x", lambda: vali.PySurfaceUD(DEV), vali.NV12, vali.RGB, (w, h), (w * 2 // 3, h * 2 // 3), lambda t, a, b: t.RunAsync(a, b), lambda t, b: t.RunBatchAsync(b))
    run("UD 2x + 90 deg in one pass", lambda: vali.PySurfaceUD(DEV), vali.NV12, vali.RGB, (w, h), (h // 2, w // 2), lambda t, a, b: t.RunRotatedAsync(a, b, 90.0), lambda t, b: t.RunRotatedBatchAsync(b, angle=90.0))
    run("rotate RGB 90", lambda: vali.PySurfaceRotator(DEV), vali.RGB, vali.RGB, (w, h), (h, w), lambda t, a, b: t.RunAsync(a, b, 90.0), lambda t, b: t.RunBatchAsync(b, angle=90.0))
    run("rotate RGB 180", lambda: vali.PySurfaceRotator(DEV), vali.RGB, vali.RGB, (w, h), (w, h), lambda t, a, b: t.RunAsync(a, b, 180.0), lambda t, b: t.RunBatchAsync(b, angle=180.0))
    run("preproc NV12->RGB_32F_PLANAR 640x384", lambda: vali.PySurfacePreprocessor(DEV), vali.NV12, vali.RGB_32F_PLANAR, (w, h), (640, 384), lambda t, a, b: t.RunAsync(a, b, cc), lambda t, b: t.RunBatchAsync(b, cc))
