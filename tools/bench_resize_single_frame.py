#!/usr/bin/env python3
"""Single-frame (RunAsync / small batch) latency of the resizer: stream time and host time per call."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill
L,Li=vali.Interpolation.LANCZOS,vali.Interpolation.LINEAR
def one(fmt, sw,sh,dw,dh, interp, batch):
    rs = vali.PySurfaceResizer(fmt, DEV, interpolation=interp)
    srcs=[vali.Surface.Make(fmt,sw,sh,DEV) for _ in range(max(batch,1))]; dsts=[vali.Surface.Make(fmt,dw,dh,DEV) for _ in range(max(batch,1))]
    fill(srcs)
    if batch:
        b=rs.PrepareBatch(srcs,dsts); f=lambda: rs.RunBatchAsync(b)
    else:
        f=lambda: rs.RunAsync(srcs[0], dsts[0])
    ms, wall = timed(rs.Stream, f, 200, 20)
    return round(ms*1e3,2), round(wall*1e3,2)
for name,args in (("Y lanczos RunAsync", (vali.Y,3840,2160,1920,1088,L,0)), ("Y lanczos batch1", (vali.Y,3840,2160,1920,1088,L,1)), ("Y lanczos batch2", (vali.Y,3840,2160,1920,1088,L,2)), ("Y lanczos batch4", (vali.Y,3840,2160,1920,1088,L,4)),
                  ("NV12 lanczos RunAsync", (vali.NV12,3840,2160,1920,1088,L,0)), ("NV12 linear RunAsync", (vali.NV12,3840,2160,1920,1088,Li,0)),
                  ("NV12 1080->720 lanczos RunAsync", (vali.NV12,1920,1080,1280,720,L,0)), ("NV12 1080->720 linear RunAsync", (vali.NV12,1920,1080,1280,720,Li,0))):
    print(name, "stream us, host us:", one(*args), flush=True)
