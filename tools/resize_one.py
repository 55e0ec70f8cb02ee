#!/usr/bin/env python3
"""One batched resize, for profiling: python tools/resize_one.py [lanczos|cubic|linear] SW SH DW DH [FORMAT] (batch 64, 5 timed launches; used with tools/prof_pmc.sh)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill
interp = {"lanczos": vali.Interpolation.LANCZOS, "cubic": vali.Interpolation.CUBIC, "linear": vali.Interpolation.LINEAR}[sys.argv[1] if len(sys.argv) > 1 else "lanczos"]
sw,sh,dw,dh = (int(v) for v in (sys.argv[2:6] if len(sys.argv) > 5 else (3840,2160,1920,1088)))
n=64
fmt = vali.PixelFormat[sys.argv[6]] if len(sys.argv) > 6 else vali.NV12
rs = vali.PySurfaceResizer(fmt, DEV, interpolation=interp)
srcs=[vali.Surface.Make(fmt,sw,sh,DEV) for _ in range(n)]; dsts=[vali.Surface.Make(fmt,dw,dh,DEV) for _ in range(n)]
fill(srcs); b=rs.PrepareBatch(srcs,dsts)
ms,_=timed(rs.Stream, lambda: rs.RunBatchAsync(b), 5, 1); print('us/frame', round(ms*1e3/n,3))
