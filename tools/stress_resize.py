#!/usr/bin/env python3
"""One-off stress of the resize kernels (Lanczos / bicubic / bilinear) beyond the test-suite: random formats, sizes up to ~2600, extreme
ratios (both ways), every rows-per-wave form, single surfaces and small batches; every output bit-exact vs the oracle."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import vali_amd as vali
from oracle import oracle as o

DEV = 0
FORMATS = [("NV12", np.uint8, True), ("Y", np.uint8, False), ("RGB", np.uint8, False), ("YUV420", np.uint8, True), ("P10", np.uint16, True),
           ("RGB_32F", np.float32, False), ("YUV444", np.uint8, False), ("RGB_PLANAR", np.uint8, False), ("YUV444_10bit", np.uint16, False)]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0 = time.time(); n_ok = 0
up, down = vali.PyFrameUploader(DEV), vali.PySurfaceDownloader(DEV)
while time.time() - t0 < budget:
    name, dt, even = FORMATS[rng.integers(len(FORMATS))]
    kind = rng.integers(9)
    if kind == 0:   sw, sh, dw, dh = rng.integers(2, 200, 4)
    elif kind == 1: sw, sh = rng.integers(300, 2600), rng.integers(2, 120); dw, dh = rng.integers(2, 2600), rng.integers(2, 200)
    elif kind == 2: sw, sh = rng.integers(2, 64), rng.integers(2, 64); dw, dh = rng.integers(200, 1800), rng.integers(50, 300)      # big upscale
    elif kind == 3: sw, sh = rng.integers(1500, 2600), rng.integers(100, 300); dw, dh = rng.integers(2, 120), rng.integers(2, 60)   # big downscale (gather)
    elif kind == 4: sw, sh = 4 * int(rng.integers(1, 500)), int(rng.integers(1, 300)); dw, dh = 2 * sw, 2 * sh                     # doubled both ways: resize_up2 for one-channel planes
    elif kind == 5: dw, dh = rng.integers(4, 1300), rng.integers(2, 400); sw = 2 * dw; sh = int(dh * rng.uniform(1.0, 4.5)) + 1        # 2:1 along x, shrinking rows: columns-first x2 form
    elif kind == 7: dw, dh = 16 * int(rng.integers(1, 110)), rng.integers(2, 400); sw = dw * 3 // 2; sh = int(dh * rng.uniform(1.0, 4.5)) + 1   # 3:2 along x, shrinking rows: the uniform-weight form
    elif kind == 8: sw, sh = 2 * int(rng.integers(2, 700)), 2 * int(rng.integers(1, 200)); dw, dh = sw * 3 // 2, sh * 3 // 2               # 3:2 enlargement: the static-tap form (1, 2 channels, packed RGB)
    elif kind == 6: sw, sh = rng.integers(8, 2600), rng.integers(100, 900); dw = int(sw * rng.uniform(0.3, 1.6)) or 2; dh = int(sh / rng.uniform(1.0, 7.0)) or 2  # every slot count of the columns-first form
    else:           sw, sh = rng.integers(250, 1100), rng.integers(60, 400); dw = int(sw * rng.uniform(0.4, 2.2)) or 2; dh = int(sh * rng.uniform(0.4, 2.2)) or 2
    sw, sh, dw, dh = (int(max(2, v)) for v in (sw, sh, dw, dh))
    if even: sw, sh, dw, dh = (v // 2 * 2 for v in (sw, sh, dw, dh))
    interp, iname = [(vali.Interpolation.LANCZOS, "lanczos"), (vali.Interpolation.CUBIC, "cubic"), (vali.Interpolation.LINEAR, "linear")][rng.integers(3)]
    vali.tuning.Set("RESIZE_NO_SEPARABLE", int(rng.integers(4)))
    vali.tuning.Set("ROWS_PER_WAVE", int([0, 2, 4, 8][rng.integers(4)]))   # bilinear / point forms
    vali.tuning.Set("RESIZE_FORCE_GATHER", int(rng.integers(8) == 0))       # the direct forms
    vali.tuning.Set("RESIZE_POINT", int(rng.integers(6) != 0))              # 0: no point / x2 forms
    vali.tuning.Set("RESIZE_ROWS", int([1, 1, 1, 2, 3][rng.integers(5)]))    # planes that grow: 2 no 3:2 form, 3 the LDS-staged form instead of the register form
    pf = vali.PixelFormat[name]
    src = vali.Surface.Make(pf, sw, sh, DEV)
    nel = src.HostSize // np.dtype(dt).itemsize
    host = (rng.random(nel) * (255 if dt == np.uint8 else 1023 if dt == np.uint16 else 1.0)).astype(dt)
    if name == "P10": host = (host.astype(np.uint16) << 6).astype(np.uint16)
    assert up.Run(host.view(np.uint8), src)[0]
    nb = int(rng.integers(1, 4))
    dsts = [vali.Surface.Make(pf, dw, dh, DEV) for _ in range(nb)]
    rs = vali.PySurfaceResizer(pf, DEV, interpolation=interp)
    if nb == 1: assert rs.Run(src, dsts[0])[0]
    else: assert rs.RunBatch([src] * nb, dsts)[0]
    want = o.resize_surface(host, name, sw, sh, dw, dh, iname)
    for d in dsts:
        out = np.zeros(d.HostSize, np.uint8)
        assert down.Run(d, out)[0]
        if not np.array_equal(out, want.view(np.uint8).reshape(-1)):
            print("MISMATCH", name, sw, sh, dw, dh, iname, "rows-mode", vali.tuning.Get("RESIZE_NO_SEPARABLE"), vali.tuning.Get("ROWS_PER_WAVE"), "batch", nb, flush=True)
            sys.exit(1)
    n_ok += 1
print("stress ok:", n_ok, "cases in", round(time.time() - t0, 1), "s")
