#!/usr/bin/env python3
"""Resize throughput per pixel format and filter (batch of 32, one launch), 2160p -> 1920x1088
(non-integer vertically: an exact 2x is the point-sample shortcut for every filter) and 1080p ->
720p.  GB/s = (host size of src + dst) / time.  Catches format-specific pathologies.
Prints one JSON line per case, then the markdown table of profiles/r01_resize_formats.md."""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vali_amd as vali  # noqa: E402
from bench_configs import DEV, timed  # noqa: E402
from vali_amd._native import shim  # noqa: E402

N = 32
rows = []
for fmt in ("NV12", "YUV420", "YUV444", "RGB", "RGB_PLANAR", "Y", "P10", "RGB_32F", "RGB_32F_PLANAR"):
    for (sw, sh, dw, dh) in ((3840, 2160, 1920, 1088), (1920, 1080, 1280, 720)):
        pf = vali.PixelFormat[fmt]
        srcs = [vali.Surface.Make(pf, sw, sh, DEV) for _ in range(N)]
        dsts = [vali.Surface.Make(pf, dw, dh, DEV) for _ in range(N)]
        for s in srcs:
            for p in s._planes:
                shim.memset2d_async(DEV, p.GpuMem, p.Pitch, 77, p.Width * p.ElemSize, p.Height, 0)
        shim.stream_sync(DEV, 0)
        row = {"format": fmt, "geometry": f"{sw}x{sh}->{dw}x{dh}"}
        for name, interp in (("linear", vali.Interpolation.LINEAR), ("cubic", vali.Interpolation.CUBIC),
                             ("lanczos", vali.Interpolation.LANCZOS)):
            rs = vali.PySurfaceResizer(pf, DEV, interpolation=interp)
            b = rs.PrepareBatch(srcs, dsts)
            ms, _ = timed(rs.Stream, lambda: rs.RunBatchAsync(b), 10)
            row[name + "_us_per_frame"] = round(ms * 1e3 / N, 2)
            row[name + "_GBps"] = round((srcs[0].HostSize + dsts[0].HostSize) * N / (ms * 1e-3) / 1e9, 1)
        print(json.dumps(row), flush=True)
        rows.append(row)
        del srcs, dsts

print("| format | geometry | bilinear µs/frame | GB/s | bicubic µs/frame | GB/s | Lanczos-3 µs/frame | GB/s |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['format']} | {r['geometry']} | {r['linear_us_per_frame']} | {r['linear_GBps']} | "
          f"{r['cubic_us_per_frame']} | {r['cubic_GBps']} | {r['lanczos_us_per_frame']} | {r['lanczos_GBps']} |")
