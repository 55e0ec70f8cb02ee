#!/usr/bin/env python3
"""One batched fused UD + quarter turn (NV12 -> RGB, rotated) for profiling: python tools/udrot_one.py SW SH DW DH [ANGLE] (batch 64; with tools/prof_pmc.sh)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill
sw, sh, dw, dh = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (3840, 2160, 1920, 1080)
angle = float(sys.argv[5]) if len(sys.argv) > 5 else 90.0
n = 64
ud = vali.PySurfaceUD(DEV)
srcs = [vali.Surface.Make(vali.NV12, sw, sh, DEV) for _ in range(n)]; dsts = [vali.Surface.Make(vali.RGB, dh, dw, DEV) for _ in range(n)]
fill(srcs); b = ud.PrepareBatch(srcs, dsts)
ms, _ = timed(ud.Stream, lambda: ud.RunRotatedBatchAsync(b, angle=angle), 5, 1)
print('us/frame', round(ms * 1e3 / n, 3), 'TB/s', round((sw * sh * 1.5 + dw * dh * 3) / (ms * 1e-3 / n) / 1e12, 3))
