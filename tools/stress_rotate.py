#!/usr/bin/env python3
"""One-off stress of the rotator beyond the test-suite: random formats and sizes (incl. >= 4 Mpixel packed frames: the
tall-tile form), quarter / half turns (exact permutations) and arbitrary angles with shifts (bit-exact vs the oracle),
single surfaces and small batches.   python tools/stress_rotate.py [seed] [seconds]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import vali_amd as vali
from oracle import oracle as o

DEV = 0
FORMATS = [("RGB", np.uint8, 3, False), ("BGR", np.uint8, 3, False), ("Y", np.uint8, 1, False), ("YUV444", np.uint8, 1, False),
           ("YUV420", np.uint8, 1, True), ("RGB_32F", np.float32, 3, False), ("YUV444_10bit", np.uint16, 1, False)]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0 = time.time(); n_ok = 0
up, down = vali.PyFrameUploader(DEV), vali.PySurfaceDownloader(DEV)
rot = vali.PySurfaceRotator(DEV)
while time.time() - t0 < budget:
    name, dt, ch, even = FORMATS[rng.integers(len(FORMATS))]
    kind = rng.integers(4)
    if kind == 0:   w, h = rng.integers(1, 200, 2)
    elif kind == 1: w, h = rng.integers(200, 2200), rng.integers(1, 300)
    elif kind == 2: w, h = rng.integers(1, 300), rng.integers(200, 2200)
    else:           w, h = rng.integers(2000, 2600), rng.integers(1700, 2100)      # >= 4 Mpixel sometimes
    w, h = int(w), int(h)
    if even: w, h = max(2, w // 2 * 2), max(2, h // 2 * 2)
    if name == "RGB_32F" and w * h > 1 << 20: w, h = w // 4 + 1, h // 4 + 1
    quarter = rng.integers(3) != 0 or name in ("YUV444", "YUV420", "YUV444_10bit")   # arbitrary angles: one-plane formats
    angle = float([90.0, 180.0, 270.0, -90.0, 0.0][rng.integers(5)]) if quarter else float(rng.uniform(-180, 180))
    if not quarter and w * h > 1 << 20: w, h = w // 3 + 1, h // 3 + 1
    if even: w, h = max(2, w // 2 * 2), max(2, h // 2 * 2)
    n90 = (int(round(angle)) + 360) % 360 if quarter else 0
    dw, dh = (h, w) if n90 in (90, 270) else (w, h)
    vali.tuning.Set("ROTATE_NO_TILE", int(rng.integers(3)) if quarter else 0)
    pf = vali.PixelFormat[name]
    src = vali.Surface.Make(pf, w, h, DEV)
    nel = src.HostSize // np.dtype(dt).itemsize
    host = (rng.random(nel) * (255 if dt == np.uint8 else 1023 if dt == np.uint16 else 1.0)).astype(dt)
    assert up.Run(host.view(np.uint8), src)[0]
    nb = int(rng.integers(1, 4))
    dsts = [vali.Surface.Make(pf, dw, dh, DEV) for _ in range(nb)]
    for d in dsts:   # pixels whose source falls outside the frame are left untouched (NPP semantics): known contents
        assert up.Run(np.zeros(d.HostSize, np.uint8), d)[0]
    sx, sy = (0.0, 0.0) if quarter else (float(rng.uniform(-20, w)), float(rng.uniform(-20, h)))
    if nb == 1: ok, info = rot.Run(src, dsts[0], angle, sx, sy)
    else: ok, info = rot.RunBatch([src] * nb, dsts, angle, sx, sy)
    assert ok, (info, name, w, h, angle)
    want, off = [], 0
    for sp, dp in zip(src.Planes, dsts[0].Planes):
        pw, ph = sp.Width // ch, sp.Height
        plane = host[off: off + sp.Width * sp.Height].reshape(ph, sp.Width)
        if quarter:
            a, px, py = o.canonical_shifts(angle, pw, ph)
        else:
            a, px, py = angle, sx / (w // pw), sy / (h // ph)     # chroma planes: shifts scale with the plane
        want.append(o.rotate_plane(np.ascontiguousarray(plane), ch, dp.Width // ch, dp.Height, a, px, py).reshape(-1))
        off += sp.Width * sp.Height
    want = np.concatenate(want).view(np.uint8)
    for d in dsts:
        got = np.zeros(d.HostSize, np.uint8)
        assert down.Run(d, got)[0]
        if not np.array_equal(got, want):
            print("MISMATCH", name, w, h, angle, sx, sy, "batch", nb, "no_tile", vali.tuning.Get("ROTATE_NO_TILE"), flush=True)
            sys.exit(1)
    n_ok += 1
print("stress ok:", n_ok, "cases in", round(time.time() - t0, 1), "s")
