import sys, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import vali_amd as vali
import bench_configs as bc
from bench_configs import DEV, timed, fill
def run(sw,sh,dw,dh,interp,n=32):
    rs = vali.PySurfaceResizer(vali.NV12, DEV, interpolation=interp)
    srcs=[vali.Surface.Make(vali.NV12,sw,sh,DEV) for _ in range(n)]; dsts=[vali.Surface.Make(vali.NV12,dw,dh,DEV) for _ in range(n)]
    fill(srcs); b=rs.PrepareBatch(srcs,dsts)
    ms,_=timed(rs.Stream, lambda: rs.RunBatchAsync(b), 20); return round(ms*1e3/n,3)
for var in (0,1,2,3,4,7):
    vali.tuning.Set("RESIZE_NO_SEPARABLE", var)
    print(var, 'lanczos 2160->1088', run(3840,2160,1920,1088,vali.Interpolation.LANCZOS), 'cubic', run(3840,2160,1920,1088,vali.Interpolation.CUBIC), 'lanczos up', run(1920,1080,3840,2160,vali.Interpolation.LANCZOS,16), flush=True)
