#!/usr/bin/env python3
"""One batched UD (NV12 -> dst format) for profiling: python tools/ud_one.py SW SH DW DH [FORMAT] (batch 64; used with tools/prof_pmc.sh)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill
sw, sh, dw, dh = (int(v) for v in sys.argv[1:5])
fmt = vali.PixelFormat[sys.argv[5]] if len(sys.argv) > 5 else vali.RGB
n = 64
ud = vali.PySurfaceUD(DEV)
srcs = [vali.Surface.Make(vali.NV12, sw, sh, DEV) for _ in range(n)]; dsts = [vali.Surface.Make(fmt, dw, dh, DEV) for _ in range(n)]
fill(srcs); b = ud.PrepareBatch(srcs, dsts)
ms, _ = timed(ud.Stream, lambda: ud.RunBatchAsync(b), 5, 1)
byts = sw * sh * 3 // 2 + sum(d for d in [dsts[0].HostSize])
print('us/frame', round(ms * 1e3 / n, 3), 'TB/s', round(byts / (ms * 1e-3 / n) / 1e12, 3))
