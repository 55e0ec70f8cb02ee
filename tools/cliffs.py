#!/usr/bin/env python3
"""Geometry cliffs: every batched operator at a round geometry and at ragged neighbours of it; prints us/frame and the
throughput (source + destination bytes per us) relative to the round geometry.  Anything far below 1.0 is a path that
fell off the vector forms.   python tools/cliffs.py [op ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill

N = 32


def run(make_task, fmt_s, fmt_d, s, d, call):
    task = make_task()
    srcs = [vali.Surface.Make(fmt_s, s[0], s[1], DEV) for _ in range(N)]
    dsts = [vali.Surface.Make(fmt_d, d[0], d[1], DEV) for _ in range(N)]
    fill(srcs)
    b = task.PrepareBatch(srcs, dsts)
    ms, _ = timed(task.Stream, lambda: call(task, b), 10, 2, 0.05)
    return ms * 1e3 / N, (srcs[0].HostSize + dsts[0].HostSize) / (ms * 1e3 / N)   # us, bytes per us


def even(v):
    return v // 2 * 2


OPS = {}
cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
for name, fd in (("cvt_nv12_rgb", vali.RGB), ("cvt_nv12_planar", vali.RGB_PLANAR)):
    OPS[name] = [(lambda: vali.PySurfaceConverter(DEV), vali.NV12, fd, (w, h), (w, h), lambda t, b: t.RunBatchAsync(b, cc))
                 for (w, h) in ((1920, 1080), (1918, 1078), (1366, 768), (1914, 1082))]
for a, b in (("RGB", "RGB_PLANAR"), ("RGB_PLANAR", "RGB"), ("RGB", "YUV420"), ("YUV420", "RGB"), ("YUV420", "NV12"), ("NV12", "YUV420"),
             ("RGB", "BGR"), ("RGB", "YUV444"), ("RGB_PLANAR", "YUV444"), ("P10", "NV12"), ("RGB", "RGB_32F"), ("RGB_32F", "RGB_32F_PLANAR")):
    OPS[f"cvt_{a}_{b}"] = [(lambda: vali.PySurfaceConverter(DEV), vali.PixelFormat[a], vali.PixelFormat[b], (w, h), (w, h),
                            lambda t, bt: t.RunBatchAsync(bt, cc)) for (w, h) in ((1920, 1080), (1918, 1078), (1366, 768))]
for filt in ("LINEAR", "LANCZOS"):
    it = vali.Interpolation[filt]
    OPS["resize_" + filt.lower()] = [(lambda it=it: vali.PySurfaceResizer(vali.NV12, DEV, interpolation=it), vali.NV12, vali.NV12, s, d,
                                      lambda t, b: t.RunBatchAsync(b))
                                     for (s, d) in (((1920, 1080), (1280, 720)), ((1918, 1078), (1278, 718)), ((1920, 1080), (1274, 714)), ((1366, 768), (854, 480)))]
    OPS["resize_rgb_" + filt.lower()] = [(lambda it=it: vali.PySurfaceResizer(vali.RGB, DEV, interpolation=it), vali.RGB, vali.RGB, s, d,
                                          lambda t, b: t.RunBatchAsync(b))
                                         for (s, d) in (((1920, 1080), (1280, 720)), ((1918, 1078), (1277, 719)), ((1366, 768), (853, 481)))]
for ang in (90.0, 180.0, 270.0):
    for f in ("RGB", "Y", "YUV420"):
        pf = vali.PixelFormat[f]
        geos = ((1920, 1080), (1918, 1078), (1366, 768)) if f == "YUV420" else ((1920, 1080), (1918, 1078), (1917, 1079), (1366, 768))
        OPS[f"rotate_{f}_{int(ang)}"] = [(lambda: vali.PySurfaceRotator(DEV), pf, pf, (w, h), ((w, h) if ang == 180.0 else (h, w)),
                                          lambda t, b, ang=ang: t.RunBatchAsync(b, angle=ang)) for (w, h) in geos]
for fd in ("RGB", "RGB_PLANAR", "YUV444", "RGB_32F_PLANAR"):
    OPS["ud_" + fd] = [(lambda: vali.PySurfaceUD(DEV), vali.NV12, vali.PixelFormat[fd], s, d, lambda t, b: t.RunBatchAsync(b))
                       for (s, d) in (((1920, 1080), (960, 540)), ((1916, 1076), (958, 538)), ((1920, 1080), (1280, 720)), ((1918, 1078), (1277, 717)),
                                      ((1920, 1080), (1920, 1080)), ((1918, 1078), (1918, 1078)))]
OPS["ud_P10_YUV444_10bit"] = [(lambda: vali.PySurfaceUD(DEV), vali.P10, vali.YUV444_10bit, s, d, lambda t, b: t.RunBatchAsync(b))
                              for (s, d) in (((1920, 1080), (960, 540)), ((1920, 1080), (1280, 720)), ((1918, 1078), (1277, 717)))]
OPS["ud_planar_YUV420_YUV444"] = [(lambda: vali.PySurfaceUD(DEV), vali.YUV420, vali.YUV444, s, d, lambda t, b: t.RunBatchAsync(b))
                                  for (s, d) in (((1920, 1080), (960, 540)), ((1920, 1080), (1280, 720)), ((1918, 1078), (1277, 717)))]
for out in ("RGB_32F_PLANAR", "RGB"):
    OPS["preproc_" + out] = [(lambda: vali.PySurfacePreprocessor(DEV), vali.NV12, vali.PixelFormat[out], s, d, lambda t, b: t.RunBatchAsync(b, cc))
                             for (s, d) in (((1920, 1080), (1920, 1080)), ((1918, 1078), (1918, 1078)), ((1920, 1080), (640, 384)), ((1918, 1078), (638, 382)))]
for ang in (270.0,):   # heights that are not multiples of 4 mirror the transposed segments onto odd offsets
    OPS[f"ud_rot_{int(ang)}"] = [(lambda: vali.PySurfaceUD(DEV), vali.NV12, vali.RGB, s, (d[1], d[0]),
                                  lambda t, b, ang=ang: t.RunRotatedBatchAsync(b, angle=ang))
                                 for (s, d) in (((1920, 1080), (960, 540)), ((1920, 1084), (960, 542)), ((1920, 1080), (1280, 718)))]
for ang in (90.0, 180.0):
    OPS[f"ud_rot_{int(ang)}"] = [(lambda: vali.PySurfaceUD(DEV), vali.NV12, vali.RGB, s, ((d[1], d[0]) if ang == 90.0 else d),
                                  lambda t, b, ang=ang: t.RunRotatedBatchAsync(b, angle=ang))
                                 for (s, d) in (((1920, 1080), (960, 540)), ((1916, 1076), (958, 538)), ((1684, 466), (842, 233)))]

if __name__ == "__main__":
    for op in (sys.argv[1:] or OPS):
        base = None
        for (mk, fs, fd, s, d, call) in OPS[op]:
            us, rate = run(mk, fs, fd, s, d, call)
            base = base or rate
            flag = "   <-- cliff" if rate / base < 0.7 else ""
            print(f"{op:24s} {s[0]}x{s[1]} -> {d[0]}x{d[1]}: {us:7.3f} us  {rate / 1e6:6.2f} TB/s  rel {rate / base:4.2f}{flag}", flush=True)
