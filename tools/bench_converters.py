#!/usr/bin/env python3
"""Throughput of every PySurfaceConverter pair (batched, one launch per step) on one GPU.

Algorithmic bytes per frame = host size of src + host size of dst (each byte read / written once).
Kernel time = HIP events on the converter's stream.  Usage: bench_converters.py [WxH] [batch]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vali_amd as vali  # noqa: E402
from vali_amd._native import shim  # noqa: E402

DEV = 0
W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3840x2160").split("x"))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 48


def main():
    cvt = vali.PySurfaceConverter(DEV)
    up = vali.PyFrameUploader(DEV)
    rows = []
    import os
    only = os.environ.get("PAIRS")
    for sf, df in vali.PySurfaceConverter.Conversions():
        if only and f"{sf.name}->{df.name}" not in only.split(","):
            continue
        srcs = [vali.Surface.Make(sf, W, H, DEV) for _ in range(N)]
        dsts = [vali.Surface.Make(df, W, H, DEV) for _ in range(N)]
        host = np.random.default_rng(0).integers(16, 236, srcs[0].HostSize, dtype=np.uint8)
        assert up.Run(host, srcs[0])[0]
        for s in srcs[1:]:
            for a, b in zip(srcs[0]._planes, s._planes):
                shim.memcpy2d_async(DEV, b.GpuMem, b.Pitch, a.GpuMem, a.Pitch, a.Width * a.ElemSize, a.Height, 2, 0)
        shim.stream_sync(DEV, 0)
        batch = cvt.PrepareBatch(srcs, dsts)
        for _ in range(3):
            ok, info = cvt.RunBatchAsync(batch)
            assert ok, (sf, df, info)
        shim.stream_sync(DEV, cvt.Stream)
        a, b = shim.event_create(DEV), shim.event_create(DEV)
        reps = 10
        shim.event_record(DEV, a, cvt.Stream)
        for _ in range(reps):
            cvt.RunBatchAsync(batch)
        shim.event_record(DEV, b, cvt.Stream)
        shim.event_sync(DEV, b)
        ms = shim.event_elapsed_ms(a, b) / reps
        bytes_frame = srcs[0].HostSize + dsts[0].HostSize
        if (sf, df) == (vali.NV12, vali.Y):
            bytes_frame = 2 * W * H          # only the luma plane is read
        gbps = bytes_frame * N / (ms * 1e-3) / 1e9
        rows.append({"pair": f"{sf.name}->{df.name}", "bytes_per_frame": bytes_frame,
                     "us_per_frame": round(ms * 1e3 / N, 3), "GBps": round(gbps, 1),
                     "frac_of_8TBps": round(gbps / 8000.0, 3)})
        print(json.dumps(rows[-1]), flush=True)
        del batch, srcs, dsts


if __name__ == "__main__":
    main()
