#!/usr/bin/env python3
"""Quarter-turn rotation throughput under the two tile walk orders (ROTATE_NO_TILE 0 / 2)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill
def run(fmt, w,h, n=64, angle=90.0):
    rot = vali.PySurfaceRotator(DEV)
    srcs=[vali.Surface.Make(fmt,w,h,DEV) for _ in range(n)]; dsts=[vali.Surface.Make(fmt,h,w,DEV) for _ in range(n)]
    fill(srcs); b=rot.PrepareBatch(srcs,dsts)
    ms,_=timed(rot.Stream, lambda: rot.RunBatchAsync(b, angle=angle), 30, 3)
    px = {vali.RGB:3, vali.Y:1, vali.RGB_32F:12}[fmt]
    return round(ms*1e3/n,3), round(2*w*h*px*n/(ms*1e-3)/1e9,1)
for order in (0,2,0,2):
    vali.tuning.Set("ROTATE_NO_TILE", order)
    print('order', order, 'RGB 1080p', run(vali.RGB,1920,1080), 'RGB 2160p', run(vali.RGB,3840,2160,32), 'Y 1080p', run(vali.Y,1920,1080), 'RGB 1080p 270', run(vali.RGB,1920,1080,64,270.0), flush=True)
