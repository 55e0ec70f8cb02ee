#!/usr/bin/env python3
"""Lanczos / bicubic / bilinear throughput over formats and geometries, under the rows-per-wave forms of the taps kernel (the table of profiles/r02_lanczos.md)."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
import bench_configs as bc
from bench_configs import DEV, timed, fill
def run(fmt,sw,sh,dw,dh,interp,n=32):
    rs = vali.PySurfaceResizer(fmt, DEV, interpolation=interp)
    srcs=[vali.Surface.Make(fmt,sw,sh,DEV) for _ in range(n)]; dsts=[vali.Surface.Make(fmt,dw,dh,DEV) for _ in range(n)]
    fill(srcs); b=rs.PrepareBatch(srcs,dsts)
    ms,_=timed(rs.Stream, lambda: rs.RunBatchAsync(b), 20); return round(ms*1e3/n,3)
L,Cu,Li=vali.Interpolation.LANCZOS,vali.Interpolation.CUBIC,vali.Interpolation.LINEAR
for small in (0,1):
    vali.tuning.Set("RESIZE_NO_SEPARABLE", small)
    print('8-row waves only' if small else 'rows per wave by launch size')
    print('  NV12 2160->1088: lanczos', run(vali.NV12,3840,2160,1920,1088,L,64), 'cubic', run(vali.NV12,3840,2160,1920,1088,Cu,64), 'linear', run(vali.NV12,3840,2160,1920,1088,Li,64), flush=True)
    print('  NV12 2160->1936x1088 (no integer ratio on either axis): lanczos', run(vali.NV12,3840,2160,1936,1088,L,64), 'cubic', run(vali.NV12,3840,2160,1936,1088,Cu,64), flush=True)
    print('  NV12 1080->2160: lanczos', run(vali.NV12,1920,1080,3840,2160,L,16), 'cubic', run(vali.NV12,1920,1080,3840,2160,Cu,16), flush=True)
    print('  Y 2160->1088 lanczos', run(vali.Y,3840,2160,1920,1088,L,64), ' YUV420', run(vali.YUV420,3840,2160,1920,1088,L,64), flush=True)
    print('  RGB 2160->1080(x1.99) lanczos', run(vali.RGB,3840,2160,1930,1086,L,32), ' RGB_32F 2160->1080 lanczos', run(vali.RGB_32F,3840,2160,1920,1080,L,8), 'RGB_32F x1.99', run(vali.RGB_32F,3840,2160,1930,1086,L,8), flush=True)
    print('  NV12 single frame 2160->1088 lanczos', run(vali.NV12,3840,2160,1920,1088,L,1), flush=True)
