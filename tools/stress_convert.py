#!/usr/bin/env python3
"""One-off stress of the converters' ragged / misaligned path: random pair, size, base offsets and pitches through the C ABI,
bit-exact vs the oracle and no byte written outside the destination rows (helpers of tests/test_gpu_ragged.py)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import vali_amd as vali
from vali_amd._native import shim
from vali_amd import tasks
from oracle import oracle as o
import test_gpu_ragged as tr

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
pairs = [(s.name, d.name) for s, d in vali.PySurfaceConverter.Conversions()]
t0 = time.time(); n = 0
while time.time() - t0 < budget:
    src, dst = pairs[rng.integers(len(pairs))]
    sub = any(f in ("NV12", "YUV420", "P10", "P12") for f in (src, dst))
    w = int(rng.integers(1, 1400)); h = int(rng.integers(1, 40))
    if rng.random() < 0.3: w = int(rng.integers(1, 5)) * 16 + int(rng.integers(-3, 4))
    w, h = max(w, 1), max(h, 1)
    if sub: w, h = max(2, w // 2 * 2), max(2, h // 2 * 2)
    mis, extra = int(rng.integers(0, 16)), int(rng.integers(0, 9))
    host = tr.host_image(src, w, h, int(rng.integers(1 << 30)), o)
    arena = tr.Arena(vali, 0, 2 * (o.host_size(src, w, h) + o.host_size(dst, w, h)) + (1 << 16) + 64 * h * 8 + 4096)
    try:
        sd, _ = tr.place(arena, o, src, w, h, host, mis, extra)
        dd, dplanes = tr.place(arena, o, dst, w, h, None, int(rng.integers(0, 16)), int(rng.integers(0, 9)))
        sp, op = tr._params(vali, o, src, dst)
        if src == "NV12" and dst in ("RGB", "BGR", "RGB_PLANAR"):
            rc = shim.nv12_to_rgb(sd, dd, tasks._csc(tasks.CSC_NPP_709HDTV), 0)
        else:
            rc = shim.convert(sd, dd, sp, 0)
        assert rc == 0, shim.last_error()
        shim.stream_sync(0, 0)
        buf = arena.download_all()
        want = o.convert(host, src, dst, w, h, op)
        if not np.array_equal(tr.gather(buf, dplanes), want):
            print("MISMATCH", src, dst, w, h, mis, extra); sys.exit(1)
        tr.check_sentinel(buf, arena, dplanes)
    finally:
        arena.free()
    n += 1
print("stress ok:", n, "cases in", round(time.time() - t0, 1), "s")
