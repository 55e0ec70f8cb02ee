"""CPU checks of the generic-converter restatement (oracle/vali_oracle_cvt.c): structural
identities that hold regardless of the (unpinnable) NPP rounding details, and agreement of
the product's coefficient tables with the oracle's."""
import numpy as np
import pytest

from vali_amd import tasks


def rnd(n, seed=0):
    return np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8)


def test_rgb2yuv_tables_match_product(oracle):
    assert np.array_equal(np.array(oracle.rgb2yuv_rows(0), np.float32), np.array(tasks.RGB2YUV_NPP_YUV, np.float32))
    assert np.array_equal(np.array(oracle.rgb2yuv_rows(1), np.float32), np.array(tasks.RGB2YUV_NPP_YCBCR, np.float32))


def test_shuffles_are_exact_inverses(oracle):
    w, h = 48, 20
    p = oracle.cvt_params()
    rgb = rnd(w * h * 3)
    planar = oracle.convert(rgb, "RGB", "RGB_PLANAR", w, h, p)
    assert np.array_equal(planar.reshape(3, h, w), rgb.reshape(h, w, 3).transpose(2, 0, 1))
    assert np.array_equal(oracle.convert(planar, "RGB_PLANAR", "RGB", w, h, p), rgb)
    bgr = oracle.convert(rgb, "RGB", "BGR", w, h, p)
    assert np.array_equal(bgr.reshape(h, w, 3), rgb.reshape(h, w, 3)[..., ::-1])
    nv12 = rnd(w * h * 3 // 2, 1)
    yuv420 = oracle.convert(nv12, "NV12", "YUV420", w, h, p)
    assert np.array_equal(yuv420[: w * h], nv12[: w * h])
    uv = nv12[w * h:].reshape(h // 2, w // 2, 2)
    assert np.array_equal(yuv420[w * h:].reshape(2, h // 2, w // 2), uv.transpose(2, 0, 1))
    assert np.array_equal(oracle.convert(yuv420, "YUV420", "NV12", w, h, p), nv12)
    assert np.array_equal(oracle.convert(nv12, "NV12", "Y", w, h, p), nv12[: w * h])
    y444 = oracle.convert(nv12[: w * h], "Y", "YUV444", w, h, p)
    assert np.array_equal(y444[: w * h], nv12[: w * h]) and (y444[w * h:] == 128).all()


def test_yuv420_rgb_equals_nv12_rgb(oracle):
    """Planar and semi-planar 4:2:0 give the same RGB (same NPP model, nearest chroma)."""
    w, h = 64, 32
    nv12 = rnd(w * h * 3 // 2, 2)
    yuv420 = oracle.convert(nv12, "NV12", "YUV420", w, h, oracle.cvt_params())
    for variant in (0, 3):
        p = oracle.cvt_params(csc_variant=variant)
        a = oracle.convert(nv12, "NV12", "RGB", w, h, p)
        assert np.array_equal(a, oracle.convert(yuv420, "YUV420", "RGB", w, h, p))
        assert np.array_equal(a, oracle.nv12_to_rgb(nv12.reshape(h * 3 // 2, w), w, h, oracle.csc(variant)).reshape(-1))
        b = oracle.convert(yuv420, "YUV420", "BGR", w, h, p)
        assert np.array_equal(a.reshape(h, w, 3)[..., ::-1], b.reshape(h, w, 3))


def test_rgb_yuv_matrices_against_float64(oracle):
    w, h = 40, 16
    rgb = rnd(w * h * 3, 3)
    R, G, B = (rgb.reshape(h, w, 3)[..., i].astype(np.float64) for i in range(3))
    for variant, rows in ((0, tasks.RGB2YUV_NPP_YUV), (1, tasks.RGB2YUV_NPP_YCBCR)):
        out = oracle.convert(rgb, "RGB", "YUV444", w, h, oracle.cvt_params(rgb2yuv_variant=variant))
        for c in range(3):
            kr, kg, kb, off = rows[c]
            ref = np.clip(np.rint(kr * R + kg * G + kb * B + off), 0, 255)
            assert np.abs(out.reshape(3, h, w)[c].astype(np.float64) - ref).max() <= 1
    gray = oracle.convert(rgb, "RGB", "Y", w, h, oracle.cvt_params(rgb2yuv_variant=0))
    assert np.abs(gray.reshape(h, w) - np.rint(0.299 * R + 0.587 * G + 0.114 * B)).max() <= 1
    # 4:2:0: luma identical to the 4:4:4 luma, chroma = 2x2 mean within 1 LSB
    y420 = oracle.convert(rgb, "RGB", "YUV420", w, h, oracle.cvt_params(rgb2yuv_variant=0))
    y444 = oracle.convert(rgb, "RGB", "YUV444", w, h, oracle.cvt_params(rgb2yuv_variant=0))
    assert np.array_equal(y420[: w * h], y444[: w * h])
    u = -0.147 * R - 0.289 * G + 0.436 * B + 128
    um = u.reshape(h // 2, 2, w // 2, 2).mean(axis=(1, 3))
    assert np.abs(y420[w * h: w * h + w * h // 4].reshape(h // 2, w // 2) - np.clip(np.rint(um), 0, 255)).max() <= 1


def test_element_type_pairs(oracle):
    w, h = 32, 8
    p = oracle.cvt_params()
    p10 = (np.random.default_rng(4).integers(0, 1024, w * h * 3 // 2, dtype=np.uint16) << 6).astype(np.uint16)
    nv12 = oracle.convert(p10.view(np.uint8), "P10", "NV12", w, h, p)
    assert np.array_equal(nv12, np.minimum((p10.astype(np.uint32) + 128) >> 8, 255).astype(np.uint8))
    rgb = rnd(w * h * 3, 5)
    f = oracle.convert(rgb, "RGB", "RGB_32F", w, h, p).view(np.float32)
    assert np.array_equal(f, rgb.astype(np.float32) / np.float32(255.0))
    fp = oracle.convert(f.view(np.uint8), "RGB_32F", "RGB_32F_PLANAR", w, h, p).view(np.float32)
    assert np.array_equal(fp.reshape(3, h, w), f.reshape(h, w, 3).transpose(2, 0, 1))
