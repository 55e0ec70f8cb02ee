#!/usr/bin/env python3
"""One batched quarter-turn rotation for profiling: python tools/rotate_one.py [FORMAT W H ANGLE] (batch 64; with tools/prof_pmc.sh)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill
fmt = vali.PixelFormat[sys.argv[1]] if len(sys.argv) > 1 else vali.RGB
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
angle = float(sys.argv[4]) if len(sys.argv) > 4 else 90.0
n = 64
rot = vali.PySurfaceRotator(DEV)
srcs = [vali.Surface.Make(fmt, w, h, DEV) for _ in range(n)]; dsts = [vali.Surface.Make(fmt, *((w, h) if angle % 180.0 == 0.0 else (h, w)), DEV) for _ in range(n)]
fill(srcs); b = rot.PrepareBatch(srcs, dsts)
ms, _ = timed(rot.Stream, lambda: rot.RunBatchAsync(b, angle=angle), 5, 1)
print('us/frame', round(ms * 1e3 / n, 3), 'TB/s', round(2 * srcs[0].HostSize / (ms * 1e-3 / n) / 1e12, 3))
