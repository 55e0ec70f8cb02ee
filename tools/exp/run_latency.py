#!/usr/bin/env python3
"""Where the time of a blocking Run goes (BASELINE config 2): host cost of RunAsync, of an idle stream_wait / stream_sync, Run on a tiny and on a 1080p surface."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import vali_amd as vali
from vali_amd._native import shim
def per_call(fn, n=3000):
    for _ in range(200): fn()
    shim.stream_sync(0, cvt.Stream)
    t0 = time.perf_counter()
    for _ in range(n): fn()
    shim.stream_sync(0, cvt.Stream)
    return (time.perf_counter() - t0) / n * 1e6
cvt = vali.PySurfaceConverter(0)
cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
for w, h in ((64, 48), (1920, 1080)):
    src = vali.Surface.Make(vali.NV12, w, h, 0); dst = vali.Surface.Make(vali.RGB_PLANAR, w, h, 0)
    for mode in (0, 1):
        vali.tuning.Set("BLOCKING_WAIT", mode)
        print(f"{w}x{h} wait={'word' if mode == 0 else 'hipStreamSynchronize'}: RunAsync {per_call(lambda: cvt.RunAsync(src, dst, cc)):.2f} us  "
              f"Run {per_call(lambda: cvt.Run(src, dst, cc)):.2f} us  idle stream_wait {per_call(lambda: shim.stream_wait(0, cvt.Stream)):.2f} us", flush=True)
