#!/usr/bin/env python3
"""Lanczos at the ratios pipelines really use (batch 64, rotating sets are NOT used here: quick A/B numbers)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill
def run(fmt, sw, sh, dw, dh, n=64):
    rs = vali.PySurfaceResizer(fmt, DEV, interpolation=vali.Interpolation.LANCZOS)
    srcs = [vali.Surface.Make(fmt, sw, sh, DEV) for _ in range(n)]; dsts = [vali.Surface.Make(fmt, dw, dh, DEV) for _ in range(n)]
    fill(srcs); b = rs.PrepareBatch(srcs, dsts)
    ms, _ = timed(rs.Stream, lambda: rs.RunBatchAsync(b), 20)
    byts = srcs[0].HostSize + dsts[0].HostSize
    return round(ms * 1e3 / n, 3), round(byts * n / (ms * 1e-3) / 8e12, 3)
for g in ((1920, 1080, 1280, 720), (1920, 1080, 854, 480), (3840, 2160, 1920, 1088), (3840, 2160, 1936, 1088), (3840, 2160, 2560, 1440),
          (2560, 1440, 1920, 1080), (1280, 720, 1920, 1080), (1920, 1080, 640, 384)):
    print("NV12 %dx%d -> %dx%d: %s us/frame, frac %s" % (*g, *run(vali.NV12, *g)), flush=True)
print("RGB 1920x1080 -> 1280x720:", run(vali.RGB, 1920, 1080, 1280, 720))
print("YUV420 1920x1080 -> 1280x720:", run(vali.YUV420, 1920, 1080, 1280, 720))
