#!/usr/bin/env python3
"""Integer-factor (point sample) resize: the strided-window kernel (RESIZE_POINT=1) vs the staged point form (2)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill
def run(fmt,sw,sh,dw,dh,n=64):
    rs = vali.PySurfaceResizer(fmt, DEV, interpolation=vali.Interpolation.LINEAR)
    srcs=[vali.Surface.Make(fmt,sw,sh,DEV) for _ in range(n)]; dsts=[vali.Surface.Make(fmt,dw,dh,DEV) for _ in range(n)]
    fill(srcs); b=rs.PrepareBatch(srcs,dsts)
    ms,_=timed(rs.Stream, lambda: rs.RunBatchAsync(b), 50, 5); return round(ms*1e3/n,3)
for mode in (1,2,1,2):
    vali.tuning.Set("RESIZE_POINT", mode)
    print('mode', mode, 'cfg3 NV12 2160p->720p', run(vali.NV12,3840,2160,1280,720), ' 2160p->1080p', run(vali.NV12,3840,2160,1920,1080), ' 2160p->960x540', run(vali.NV12,3840,2160,960,540), ' Y 2160->720', run(vali.Y,3840,2160,1280,720), ' 1080p->540p', run(vali.NV12,1920,1080,960,540,256), flush=True)
