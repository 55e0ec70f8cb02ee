#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary when the caller hands HOST buffers (never bench.py's `value`):
PyFrameUploader.Run (pageable numpy -> pitched surface, blocking, like the reference's
CudaUploadFrame::Run) + NV12->RGB + PySurfaceDownloader.Run, per 2160p frame."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vali_amd as vali  # noqa: E402

DEV, W, H, N = 0, 3840, 2160, 40
rng = np.random.default_rng(0)
host = rng.integers(0, 256, W * H * 3 // 2, dtype=np.uint8)
out = np.zeros(W * H * 3, np.uint8)
src = vali.Surface.Make(vali.NV12, W, H, DEV)
dst = vali.Surface.Make(vali.RGB, W, H, DEV)
up, dn, cvt = vali.PyFrameUploader(DEV), vali.PySurfaceDownloader(DEV), vali.PySurfaceConverter(DEV)
for _ in range(3):
    up.Run(host, src); cvt.Run(src, dst); dn.Run(dst, out)
t0 = time.perf_counter()
for _ in range(N):
    up.Run(host, src)
t1 = time.perf_counter()
for _ in range(N):
    cvt.Run(src, dst)
t2 = time.perf_counter()
for _ in range(N):
    dn.Run(dst, out)
t3 = time.perf_counter()
u, c, d = (t1 - t0) / N, (t2 - t1) / N, (t3 - t2) / N
print(json.dumps({"config": "host ndarray -> upload -> NV12->RGB -> download, 2160p, one frame at a time (pageable memory)",
                  "upload_ms": round(u * 1e3, 3), "upload_GBps": round(host.nbytes / u / 1e9, 2),
                  "convert_ms(sync Run)": round(c * 1e3, 4),
                  "download_ms": round(d * 1e3, 3), "download_GBps": round(out.nbytes / d / 1e9, 2),
                  "frames_per_s_upload_plus_convert": round(1 / (u + c), 1),
                  "frames_per_s_full_round_trip": round(1 / (u + c + d), 1)}))
