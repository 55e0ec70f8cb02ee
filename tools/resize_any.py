#!/usr/bin/env python3
"""One batched resize on ROTATING surface sets (>= 1.5 GiB per timed loop, as bench.py's secondary entries):
python tools/resize_any.py [lanczos|cubic|linear] SW SH DW DH [FORMAT]   (batch 64; with tools/prof_pmc.sh / tools/exp/ab.sh)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill, sets_needed, make_sets
interp = {"lanczos": vali.Interpolation.LANCZOS, "cubic": vali.Interpolation.CUBIC, "linear": vali.Interpolation.LINEAR}[sys.argv[1] if len(sys.argv) > 1 else "lanczos"]
sw, sh, dw, dh = (int(v) for v in (sys.argv[2:6] if len(sys.argv) > 5 else (3840, 2160, 1936, 1088)))
fmt = vali.PixelFormat[sys.argv[6]] if len(sys.argv) > 6 else vali.NV12
n = 64
rs = vali.PySurfaceResizer(fmt, DEV, interpolation=interp)
size = vali.Surface.Make(fmt, sw, sh, DEV).HostSize + vali.Surface.Make(fmt, dw, dh, DEV).HostSize
def make():
    srcs = [vali.Surface.Make(fmt, sw, sh, DEV) for _ in range(n)]; dsts = [vali.Surface.Make(fmt, dw, dh, DEV) for _ in range(n)]
    fill(srcs)
    return srcs, dsts, rs.PrepareBatch(srcs, dsts)
sets = make_sets(sets_needed(size * n), make)
ms, _ = timed(rs.Stream, [lambda q=q: rs.RunBatchAsync(q) for _, _, q in sets], 18, 1)
print('us/frame', round(ms * 1e3 / n, 3), 'frac of 8 TB/s', round(size / (ms * 1e-3 / n) / 8e12, 3))
