#!/usr/bin/env python3
"""python tools/exp/resize_n.py N [lanczos SW SH DW DH]: one batched resize of N frames repeated (is the kernel faster when its source fits the Infinity Cache?)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill
n = int(sys.argv[1])
sw, sh, dw, dh = (int(v) for v in (sys.argv[3:7] if len(sys.argv) > 6 else (3840, 2160, 1936, 1088)))
rs = vali.PySurfaceResizer(vali.NV12, DEV, interpolation=vali.Interpolation.LANCZOS)
srcs = [vali.Surface.Make(vali.NV12, sw, sh, DEV) for _ in range(n)]; dsts = [vali.Surface.Make(vali.NV12, dw, dh, DEV) for _ in range(n)]
fill(srcs); b = rs.PrepareBatch(srcs, dsts)
ms, _ = timed(rs.Stream, lambda: rs.RunBatchAsync(b), 20, 3)
print('n', n, 'us/frame', round(ms * 1e3 / n, 3))
