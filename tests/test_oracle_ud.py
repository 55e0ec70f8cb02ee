"""Pins the UD restatement (oracle/vali_oracle.c: vali_oracle_ud_*) against the reference's
own golden files (tests/golden/ud_640x360_*_rows120.npz, cropped from
reference tests/data/640x360_PixelFormat.*.raw by tests/golden/make_ud_goldens.py).

The goldens' input frame needs a video decoder; two kinds of pins remain.
(1) Pixel-wise identities BETWEEN goldens, evaluated with the oracle's own stage functions:
  * u8 store stage:  RGB == oracle_store_u8(RGB_32F)           (exact, every pixel)
  * layouts:         RGB_PLANAR == transpose(RGB), RGB_32F_PLANAR == transpose(RGB_32F)
  * colour matrix:   oracle_rgb_from_yuv(bin centre of YUV444) within the quantisation bound
                     of RGB_32F and unbiased (mean error ~ 0)
(2) The SAMPLING GEOMETRY of the texture kernels (ResizeUtils.cu:36-37,68-69: X = x / s, no half-pixel
centring), end to end: the oracle fed with frame 0 re-derived from the reference's frame_0.jpg against the
640x360 NV12 -> YUV444 / RGB goldens -- 47.5 dB luma through the JPEG noise, against 23.8 / 25.9 dB for the two
conventional grids (round 4; found by the round-3 review).  The 8-bit weight rounding of the texture unit stays
unpinned: it moves a sample by < 0.5 LSB, far below the noise floor of this comparison.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import GOLDEN, frame0_nv12, psnr_offset_removed


@pytest.fixture(scope="module")
def stages(oracle):
    L = oracle.lib()
    L.vali_oracle_ud_store_u8.restype = C.c_uint8
    L.vali_oracle_ud_store_u8.argtypes = [C.c_float]
    L.vali_oracle_ud_rgb_from_yuv.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    L.vali_oracle_ud_rgb_from_yuv.restype = None
    return L


def test_u8_store_stage_matches_golden_exactly(stages):
    g = np.load(GOLDEN / "ud_640x360_nv12_rows120.npz")
    f = g["rgb_32f"].ravel()
    want = g["rgb"].ravel()
    # every distinct float of the golden through the oracle's store stage
    uniq, inv = np.unique(f, return_inverse=True)
    got_u = np.array([stages.vali_oracle_ud_store_u8(float(v)) for v in uniq], np.uint8)
    assert np.array_equal(got_u[inv], want)
    assert f.min() < 0 < 1 and want.min() == 0           # negative floats really occur -> saturate to 0


def test_planar_and_packed_goldens_are_the_same_pixels():
    g = np.load(GOLDEN / "ud_640x360_nv12_rows120.npz")
    assert np.array_equal(g["rgb_planar"], g["rgb"].transpose(2, 0, 1))
    assert np.array_equal(g["rgb_32f_planar"], g["rgb_32f"].transpose(2, 0, 1))
    p = np.load(GOLDEN / "ud_640x360_p10_rows120.npz")
    assert np.array_equal(p["rgb_32f_planar"], p["rgb_32f"].transpose(2, 0, 1))


@pytest.mark.parametrize("which,depth", [("nv12", 8), ("p10", 16)])
def test_colour_matrix_consistent_with_goldens(stages, which, depth):
    g = np.load(GOLDEN / f"ud_640x360_{which}_rows120.npz")
    yuv = g["yuv444" if depth == 8 else "yuv444_10bit"].astype(np.float64)
    scale = 2.0 ** depth
    centre = ((yuv + 0.5) / scale).astype(np.float32)      # YUV stage stores trunc(val * 2^bits)
    f = g["rgb_32f"]
    out = np.zeros(3, np.float32)
    buf = out.ctypes.data_as(C.POINTER(C.c_float))
    rows = range(0, centre.shape[1], 7)                    # subsample rows: pure-Python loop
    err = []
    for y in rows:
        for x in range(0, centre.shape[2], 3):
            stages.vali_oracle_ud_rgb_from_yuv(centre[0, y, x], centre[1, y, x], centre[2, y, x], buf)
            err.append(out - f[y, x])
    err = np.array(err, np.float64) * scale
    bound = 0.5 * np.array([1 + 1.140, 1 + 0.394 + 0.581, 1 + 2.032]) + 0.02
    assert (np.abs(err).max(axis=0) <= bound).all(), np.abs(err).max(axis=0)
    assert (np.abs(err.mean(axis=0)) < 0.06).all(), err.mean(axis=0)


def test_oracle_ud_internal_identities(oracle):
    """The same identities hold for the oracle's own full pipeline on synthetic input."""
    rng = np.random.default_rng(0)
    nv = rng.integers(0, 256, (464 * 3 // 2, 848), dtype=np.uint8)
    r = oracle.ud_nv12(nv, 848, 464, "NV12", 640, 360, "RGB")
    f = oracle.ud_nv12(nv, 848, 464, "NV12", 640, 360, "RGB_32F")
    assert np.array_equal(r, np.clip(np.trunc(f * 256.0), 0, 255).astype(np.uint8))
    rp = oracle.ud_nv12(nv, 848, 464, "NV12", 640, 360, "RGB_PLANAR")
    assert np.array_equal(rp.reshape(3, 360, 640), r.reshape(360, 640, 3).transpose(2, 0, 1))
    # point sampling is the identity for YUV: scale 2x up lands on texel centres at odd x
    y = oracle.ud_nv12(nv, 848, 464, "NV12", 1696, 928, "YUV444")
    assert np.array_equal(y[0][1::2, 1::2], nv[:464])
    # exact 2x downscale = equal-weight 2x2 mean (x=0 / y=0 clamp), SURVEY A.1
    d = oracle.ud_nv12(nv, 848, 464, "NV12", 424, 232, "YUV444")[0].astype(np.int64)
    src = nv[:464].astype(np.int64)
    mean = (src[1:-1:2, 1:-1:2] + src[2::2, 1:-1:2] + src[1:-1:2, 2::2] + src[2::2, 2::2])
    want = (mean * 16384 * 256) // (65536 * 255)           # trunc(S/(65536*255) * 256)
    assert np.abs(d[1:, 1:] - want).max() <= 1


# ---- (2) the sampling geometry of ResizeUtils.cu, end to end --------------------------------------------------------
def _bilinear(plane, X, Y):
    """float64 model of an unnormalised-coordinate linear texture fetch at (X[i], Y[j]) -- for the ALTERNATIVE grids
    only; the oracle itself is called through its C entry."""
    h, w = plane.shape
    xb, yb = X - 0.5, Y - 0.5
    i, j = np.floor(xb).astype(int), np.floor(yb).astype(int)
    a, b = xb - i, yb - j
    i0, i1, j0, j1 = np.clip(i, 0, w - 1), np.clip(i + 1, 0, w - 1), np.clip(j, 0, h - 1), np.clip(j + 1, 0, h - 1)
    p = plane.astype(np.float64)
    top = p[j0][:, i0] * (1 - a) + p[j0][:, i1] * a
    bot = p[j1][:, i0] * (1 - a) + p[j1][:, i1] * a
    return top * (1 - b)[:, None] + bot * b[:, None]


def test_texture_geometry_against_the_reference_goldens(oracle):
    """reference tests/test_PySurfaceUD.py:135-188 (NV12 848x464 -> 640x360), src/TC/src/ResizeUtils.cu:36-37,68-69."""
    pytest.importorskip("PIL.Image")
    sw, sh, dw, dh, rows = 848, 464, 640, 360, 120
    nv = frame0_nv12()
    g = np.load(GOLDEN / "ud_640x360_nv12_rows120.npz")
    yuv = oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, "YUV444")
    rgb = oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, "RGB").reshape(dh, dw, 3)
    got = {}
    for c, floor in ((0, 47.0), (1, 48.4), (2, 52.9)):                      # measured 47.55 / 48.91 / 53.41 dB
        got[c], _ = psnr_offset_removed(yuv[c][:rows], g["yuv444"][c])
        assert got[c] >= floor, (c, got[c])
    for c, floor in ((0, 46.3), (1, 46.4), (2, 41.1)):                      # measured 46.80 / 46.95 / 41.63 dB
        p, _ = psnr_offset_removed(rgb[:rows, :, c], g["rgb"][..., c])
        assert p >= floor, (c, p)
    # the frame's common luma offset (tests/test_oracle_reference_pins.py): the same -1.43 LSB against a golden that
    # involves neither NPP nor a JPEG on the golden side -> it sits in frame_0.jpg's own generation
    _, off = psnr_offset_removed(yuv[0][:rows], g["yuv444"][0])
    assert -1.8 < off < -1.1
    # every other grid loses, luma by > 20 dB
    sx, sy = np.float64(dw) / sw, np.float64(dh) / sh
    xs, ys = np.arange(dw, dtype=np.float64), np.arange(rows, dtype=np.float64)
    luma, u, v = nv[:sh], nv[sh:, 0::2], nv[sh:, 1::2]
    grids = {"texture x/s (the reference)": (xs / sx, ys / sy, xs / (2 * sx), ys / (2 * sy)),
             "centre-aligned": ((xs + 0.5) / sx, (ys + 0.5) / sy, (xs + 0.5) / (2 * sx), (ys + 0.5) / (2 * sy)),
             "pixel grid x*s": (xs / sx + 0.5, ys / sy + 0.5, xs / (2 * sx) + 0.5, ys / (2 * sy) + 0.5)}
    score = {n: [psnr_offset_removed(_bilinear(p, X if k == 0 else XC, Y if k == 0 else YC), g["yuv444"][k])[0]
                 for k, p in enumerate((luma, u, v))] for n, (X, Y, XC, YC) in grids.items()}
    mine = score["texture x/s (the reference)"]
    assert abs(mine[0] - got[0]) < 0.1                                      # the float64 model of the SAME grid agrees with the C oracle
    assert score["centre-aligned"][0] < 25.0 and score["pixel grid x*s"][0] < 27.0          # measured 23.8 / 25.9 dB
    for n in ("centre-aligned", "pixel grid x*s"):
        assert score[n][1] < mine[1] - 0.8 and score[n][2] < mine[2] - 1.2  # chroma: 49.1 / 53.9 vs 48.0 / 52.4 and 47.0 / 51.0
    # nearest chroma (no interpolation at all) loses too: 47.4 / 51.9
    def nearest(p, X, Y):
        return p[np.clip(np.floor(Y).astype(int), 0, p.shape[0] - 1)][:, np.clip(np.floor(X).astype(int), 0, p.shape[1] - 1)]
    for k, p in ((1, u), (2, v)):
        assert psnr_offset_removed(nearest(p, xs / (2 * sx), ys / (2 * sy)), g["yuv444"][k])[0] < mine[k] - 1.2
