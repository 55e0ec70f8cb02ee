#!/usr/bin/env python3
"""One batched planar UD (YUV420 -> YUV444: luma resized, chroma upsampled + resized, Lanczos like the reference's UDPlanar) for profiling:
python tools/udplanar_one.py [SW SH DW DH [10]] (batch 64; with tools/prof_pmc.sh)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import vali_amd as vali
from bench_configs import DEV, timed, fill
sw, sh, dw, dh = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (1920, 1080, 1920, 1080)))
hbd = len(sys.argv) > 5 and sys.argv[5] == "10"
sf, df = (vali.YUV420_10bit, vali.YUV444_10bit) if hbd else (vali.YUV420, vali.YUV444)
n = 64
ud = vali.PySurfaceUD(DEV)
from bench_configs import sets_needed, make_sets
def make():
    srcs = [vali.Surface.Make(sf, sw, sh, DEV) for _ in range(n)]; dsts = [vali.Surface.Make(df, dw, dh, DEV) for _ in range(n)]
    fill(srcs)
    return srcs, dsts, ud.PrepareBatch(srcs, dsts)
probe = vali.Surface.Make(sf, sw, sh, DEV).HostSize + vali.Surface.Make(df, dw, dh, DEV).HostSize
sets = make_sets(sets_needed(probe * n), make)   # >= 1.5 GiB of distinct surfaces per timed loop
ms, _ = timed(ud.Stream, [lambda q=q: ud.RunBatchAsync(q) for _, _, q in sets], 12, 1)
bytes_ = probe
print('us/frame', round(ms * 1e3 / n, 3), 'TB/s', round(bytes_ / (ms * 1e-3 / n) / 1e12, 3))
