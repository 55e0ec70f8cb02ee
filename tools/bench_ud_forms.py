#!/usr/bin/env python3
"""UD NV12 2160p -> 1080p through the exact-ratio kernels (UD_DOWN2 = 1) and the general kernel (0), per output."""
import sys
from pathlib import Path
sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else str(Path(__file__).resolve().parent.parent)); sys.path.insert(1, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill
def run(dst_fmt, sw, sh, dw, dh, n=64):
    ud = vali.PySurfaceUD(DEV)
    srcs=[vali.Surface.Make(vali.NV12,sw,sh,DEV) for _ in range(n)]; dsts=[vali.Surface.Make(dst_fmt,dw,dh,DEV) for _ in range(n)]
    fill(srcs); b=ud.PrepareBatch(srcs,dsts)
    ms,_=timed(ud.Stream, lambda: ud.RunBatchAsync(b), 30, 3); return round(ms*1e3/n,3)
for down2 in (1, 0, 1, 0):
    vali.tuning.Set("UD_DOWN2", down2)
    print('UD_DOWN2', down2, 'RGB', run(vali.RGB,3840,2160,1920,1080), 'RGB_PLANAR', run(vali.RGB_PLANAR,3840,2160,1920,1080), 'YUV444', run(vali.YUV444,3840,2160,1920,1080),
          '| 1.5x RGB', run(vali.RGB,2880,1620,1920,1080), '3x RGB', run(vali.RGB,5760,3240,1920,1080,32), flush=True)
