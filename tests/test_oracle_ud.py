"""Pins the UD restatement (oracle/vali_oracle.c: vali_oracle_ud_*) against the reference's
own golden files (tests/golden/ud_640x360_*_rows120.npz, cropped from
reference tests/data/640x360_PixelFormat.*.raw by tests/golden/make_ud_goldens.py).

The goldens' input frame needs a video decoder and is not available, so the pins are
pixel-wise identities BETWEEN goldens, evaluated with the oracle's own stage functions:
  * u8 store stage:  RGB == oracle_store_u8(RGB_32F)           (exact, every pixel)
  * layouts:         RGB_PLANAR == transpose(RGB), RGB_32F_PLANAR == transpose(RGB_32F)
  * colour matrix:   oracle_rgb_from_yuv(bin centre of YUV444) within the quantisation bound
                     of RGB_32F and unbiased (mean error ~ 0)
"""
import ctypes as C

import numpy as np
import pytest

from conftest import GOLDEN


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
