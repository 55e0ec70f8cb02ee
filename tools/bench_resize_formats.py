#!/usr/bin/env python3
"""Resize throughput per pixel format and filter (batch of 32, one launch), 2160p -> 1080p and
1080p -> 720p.  GB/s = (host size of src + dst) / time.  Catches format-specific pathologies."""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vali_amd as vali  # noqa: E402
from bench_configs import DEV, timed  # noqa: E402
from vali_amd._native import shim  # noqa: E402

N = 32
for fmt in ("NV12", "YUV420", "YUV444", "RGB", "RGB_PLANAR", "Y", "P10", "RGB_32F", "RGB_32F_PLANAR"):
    for (sw, sh, dw, dh) in ((3840, 2160, 1920, 1080), (1920, 1080, 1280, 720)):
        pf = vali.PixelFormat[fmt]
        srcs = [vali.Surface.Make(pf, sw, sh, DEV) for _ in range(N)]
        dsts = [vali.Surface.Make(pf, dw, dh, DEV) for _ in range(N)]
        for s in srcs:
            for p in s._planes:
                shim.memset2d_async(DEV, p.GpuMem, p.Pitch, 77, p.Width * p.ElemSize, p.Height, 0)
        shim.stream_sync(DEV, 0)
        row = {"format": fmt, "geometry": f"{sw}x{sh}->{dw}x{dh}"}
        for name, interp in (("linear", vali.Interpolation.LINEAR), ("lanczos", vali.Interpolation.LANCZOS)):
            rs = vali.PySurfaceResizer(pf, DEV, interpolation=interp)
            b = rs.PrepareBatch(srcs, dsts)
            ms, _ = timed(rs.Stream, lambda: rs.RunBatchAsync(b), 10)
            row[name + "_us_per_frame"] = round(ms * 1e3 / N, 2)
            row[name + "_GBps"] = round((srcs[0].HostSize + dsts[0].HostSize) * N / (ms * 1e-3) / 1e9, 1)
        print(json.dumps(row), flush=True)
        del srcs, dsts
