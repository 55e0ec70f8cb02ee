#!/usr/bin/env python3
"""One batched rotation by any angle into a destination of the SAME size (bench.py's `affine` entry for any format / size / angle), on
rotating surface sets:  python tools/rotate_any.py [FORMAT W H ANGLE [SHIFT_X SHIFT_Y]]   (batch 64; with tools/prof_pmc.sh)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill, sets_needed, make_sets
fmt = vali.PixelFormat[sys.argv[1]] if len(sys.argv) > 1 else vali.RGB
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
angle = float(sys.argv[4]) if len(sys.argv) > 4 else 30.0
sx, sy = (float(sys.argv[5]), float(sys.argv[6])) if len(sys.argv) > 6 and sys.argv[5] != "single" else (0.0, 0.0)
if "single" in sys.argv:      # one surface per call (launch-bound): stream time and host time per RunAsync
    rot = vali.PySurfaceRotator(DEV)
    s_, d_ = vali.Surface.Make(fmt, w, h, DEV), vali.Surface.Make(fmt, w, h, DEV)
    fill([s_])
    ms, wall = timed(rot.Stream, lambda: rot.RunAsync(s_, d_, angle, sx, sy), 300, 20)
    print('single call: stream us', round(ms * 1e3, 3), 'host us', round(wall * 1e3, 3))
    sys.exit(0)
n = 64 if w * h <= 1920 * 1080 else 16
rot = vali.PySurfaceRotator(DEV)
size = vali.Surface.Make(fmt, w, h, DEV).HostSize
def make():
    srcs = [vali.Surface.Make(fmt, w, h, DEV) for _ in range(n)]; dsts = [vali.Surface.Make(fmt, w, h, DEV) for _ in range(n)]
    fill(srcs)
    return srcs, dsts, rot.PrepareBatch(srcs, dsts)
sets = make_sets(sets_needed(2 * size * n), make)
ms, _ = timed(rot.Stream, [lambda q=q: rot.RunBatchAsync(q, angle=angle, shift_x=sx, shift_y=sy) for _, _, q in sets], 12, 1)
print('us/frame', round(ms * 1e3 / n, 3), 'frac of 8 TB/s (2 x pixels)', round(2 * size / (ms * 1e-3 / n) / 8e12, 3))
