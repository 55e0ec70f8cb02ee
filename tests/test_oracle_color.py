"""Corroborates the colour matrices of the restatement against the reference's own output.

tests/golden/frame_0.jpg is frame 0 of the reference's test video converted by the REFERENCE with
NV12 -> RGB (BT.709 + MPEG, i.e. nppiNV12ToRGB_709CSC) and JPEG-compressed; the UD golden
`640x360 NV12 -> YUV444` holds the same frame's YUV at 640x360.  Converting that YUV with the
oracle's matrix variants and comparing with frame_0.jpg (resampled with the UD sampling rule)
identifies the variant: 709CSC 40.5 dB, YCbCr-601 38.3 dB, both full-range variants < 30 dB.
JPEG noise bounds the PSNR, so this is corroboration (range handling + matrix family), not a
bit-level pin."""
import numpy as np
import pytest

from conftest import GOLDEN

PIL = pytest.importorskip("PIL.Image")


def ud_taps(n_dst, n_src):
    s = np.float32(n_dst) / np.float32(n_src)
    X = (np.arange(n_dst, dtype=np.float32) / s).astype(np.float64) - 0.5
    i = np.floor(X)
    a = X - i
    return np.clip(i, 0, n_src - 1).astype(int), np.clip(i + 1, 0, n_src - 1).astype(int), a


def psnr(a, b):
    return 10 * np.log10(255.0 ** 2 / np.mean((a.astype(np.float64) - b) ** 2))


def test_bt709_limited_matrix_matches_reference_frame(oracle):
    g = np.load(GOLDEN / "ud_640x360_nv12_rows120.npz")
    yuv = np.ascontiguousarray(g["yuv444"]).reshape(-1)                 # (3, 120, 640)
    img = np.asarray(PIL.open(GOLDEN / "frame_0.jpg")).astype(np.float64)   # 464 x 848 x 3
    x0, x1, ax = ud_taps(640, 848)
    y0, y1, ay = ud_taps(360, 464)

    def resample(ch):
        top = ch[y0][:, x0] * (1 - ax) + ch[y0][:, x1] * ax
        bot = ch[y1][:, x0] * (1 - ax) + ch[y1][:, x1] * ax
        return top * (1 - ay)[:, None] + bot * ay[:, None]

    ref = np.stack([resample(img[..., c]) for c in range(3)], -1)[:120]
    score = {}
    for name, variant in (("yuv_601_full", 0), ("709csc", 1), ("709hdtv", 2), ("ycbcr_601", 3)):
        rgb = oracle.convert(yuv, "YUV444", "RGB", 640, 120, oracle.cvt_params(csc_variant=variant))
        score[name] = psnr(rgb.reshape(120, 640, 3), ref)
    assert score["709csc"] >= 39.5
    assert score["709csc"] > score["ycbcr_601"] + 1.5            # matrix family
    assert score["709csc"] > score["709hdtv"] + 9.0              # limited vs full range
    assert score["709csc"] > score["yuv_601_full"] + 9.0


def test_simd_baseline_is_bit_identical_to_the_scalar_restatement():
    """oracle/vali_oracle_simd.c (the CPU baseline bench.py times) == vali_oracle_nv12_to_rgb, on
    noise, on the full-excursion gradient, for all four colour variants and ragged widths."""
    from oracle import oracle as o
    from conftest import make_nv12

    for w, h in ((64, 48), (1920, 1080), (70, 34), (10, 6), (8, 2)):
        for seed in (-1, 3):
            nv = make_nv12(w, h, seed)
            for variant in range(4):
                k = o.csc(variant)
                scalar = o.nv12_to_rgb_mt([nv], w, h, k, 1)[0]
                simd = o.nv12_to_rgb_mt([nv], w, h, k, 2, simd=True)[0]
                assert np.array_equal(scalar, simd), (w, h, seed, variant)
