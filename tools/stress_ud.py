#!/usr/bin/env python3
"""One-off stress of the UD kernels beyond the test-suite: NV12 / P10 sources, every output format, random geometries
(exact 2x and 1x widths for the k_ud_lean forms, any ratio for k_ud_nv12 staged / gather), single surfaces and small
batches, rotated outputs; every output bit-exact vs the oracle.   python tools/stress_ud.py [seed] [seconds]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import vali_amd as vali
from oracle import oracle as o

DEV = 0
OUTS = ["RGB", "RGB_PLANAR", "YUV444", "RGB_32F", "RGB_32F_PLANAR"]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0 = time.time(); n_ok = 0
up, down = vali.PyFrameUploader(DEV), vali.PySurfaceDownloader(DEV)
ud = vali.PySurfaceUD(DEV)
while time.time() - t0 < budget:
    src_name = "NV12" if rng.integers(4) else "P10"
    out = OUTS[rng.integers(len(OUTS))] if src_name == "NV12" else ["YUV444_10bit", "RGB_32F", "RGB_32F_PLANAR"][rng.integers(3)]
    kind = rng.integers(8)
    if kind == 0:   sw, sh, dw, dh = rng.integers(2, 260, 4)
    elif kind == 1: dw, dh = rng.integers(2, 1400), rng.integers(2, 300); sw, sh = 2 * dw, int(dh * rng.uniform(0.5, 3.0)) or 2   # exact 2x width
    elif kind == 2: dw, dh = rng.integers(2, 2000), rng.integers(2, 300); sw, sh = dw, int(dh * rng.uniform(0.5, 2.0)) or 2       # exact 1x width
    elif kind == 3: sw, sh = rng.integers(300, 2600), rng.integers(2, 200); dw = int(sw / rng.uniform(1.05, 3.9)) or 2; dh = int(sh / rng.uniform(0.7, 3.0)) or 2
    elif kind == 6: dw, dh = 8 * int(rng.integers(1, 200)), 4 * int(rng.integers(1, 90)); sw, sh = dw * 3 // 2, dh * 3 // 2           # exactly 3:2 both ways: k_ud_32
    elif kind == 7: dw, dh = 8 * int(rng.integers(1, 200)), 4 * int(rng.integers(1, 90)); sw, sh = 2 * dw, 2 * dh                     # exactly 2:1 both ways: k_ud_half[_t]
    elif kind == 4: sw, sh = rng.integers(1500, 4000), rng.integers(60, 200); dw, dh = rng.integers(2, 300), rng.integers(2, 60)   # gather form
    else:           sw, sh = rng.integers(2, 400), rng.integers(2, 200); dw = int(sw * rng.uniform(1.0, 3.0)) or 2; dh = int(sh * rng.uniform(1.0, 3.0)) or 2
    sw, sh = (int(max(2, v)) // 2 * 2 for v in (sw, sh))
    dw, dh = (int(max(1, v)) for v in (dw, dh))
    for k, lo in (("UD_DOWN2", 2), ("UD_FORCE_GATHER", 2), ("ROWS_PER_WAVE", 2)):
        vali.tuning.Set(k, int(rng.integers(3)) if k == "UD_DOWN2" else int([0, 2, 4, 8][rng.integers(4)]) if k == "ROWS_PER_WAVE" else int(rng.integers(lo)))
    spf, dpf = vali.PixelFormat[src_name], vali.PixelFormat[out]
    src = vali.Surface.Make(spf, sw, sh, DEV)
    dt = np.uint8 if src_name == "NV12" else np.uint16
    nel = src.HostSize // np.dtype(dt).itemsize
    host = (rng.random(nel) * (255 if dt == np.uint8 else 1023)).astype(dt)
    if src_name == "P10": host = (host.astype(np.uint16) << 6).astype(np.uint16)
    assert up.Run(host.view(np.uint8), src)[0]
    nb = int(rng.integers(1, 4))
    dsts = [vali.Surface.Make(dpf, dw, dh, DEV) for _ in range(nb)]
    if nb == 1: ok, info = ud.Run(src, dsts[0])
    else: ok, info = ud.RunBatch([src] * nb, dsts)
    assert ok, info
    want = o.ud_nv12(host.reshape(sh * 3 // 2, sw), sw, sh, src_name, dw, dh, out)
    for d in dsts:
        got = np.zeros(d.HostSize, np.uint8)
        assert down.Run(d, got)[0]
        if not np.array_equal(got, np.ascontiguousarray(want).view(np.uint8).reshape(-1)):
            print("MISMATCH", src_name, out, sw, sh, dw, dh, "batch", nb, {k: vali.tuning.Get(k) for k in ("UD_DOWN2", "UD_FORCE_GATHER", "ROWS_PER_WAVE")}, flush=True)
            sys.exit(1)
    n_ok += 1
print("stress ok:", n_ok, "cases in", round(time.time() - t0, 1), "s")
