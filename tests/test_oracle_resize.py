"""Pins the resize GEOMETRY against the reference's own fixture.

reference tests/test_PySurfaceResizer.py:64-140 resizes test.nv12 (848x464, missing) to 424x232
and compares with data/test_small.nv12 (present; 2 frames committed as
tests/golden/test_small_2frames.nv12).  Frame 0 of the same video is also available as
data/frame_0.jpg (the reference's NV12->RGB output, JPEG-compressed), so the luma of the
missing input can be re-derived to JPEG accuracy and the sampling grid of the reference's
resizer identified: src = dst * scale (top-left aligned), not the centre-aligned grid."""
import numpy as np
import pytest

from conftest import GOLDEN

PIL = pytest.importorskip("PIL.Image")


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10 * np.log10(255.0 ** 2 / mse)


@pytest.fixture(scope="module")
def luma_pair():
    rgb = np.asarray(PIL.open(GOLDEN / "frame_0.jpg")).astype(np.float64)
    y = 16 + 0.1826 * rgb[..., 0] + 0.6142 * rgb[..., 1] + 0.0620 * rgb[..., 2]   # BT.709 limited
    full = np.clip(np.rint(y), 0, 255).astype(np.uint8)                         # 848 x 464
    small = np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8)[: 424 * 232].reshape(232, 424)
    return full, small


def test_oracle_resize_lands_on_the_reference_grid(oracle, luma_pair):
    full, small = luma_pair
    got = oracle.resize_plane(np.ascontiguousarray(full), 1, 424, 232)
    assert np.array_equal(got, full[0::2, 0::2])          # integer factor -> exact point sample
    assert psnr(got, small) >= 40.0                        # 41.7 dB: JPEG + colour round trip noise
    # the alternatives are far away, so the grid is identified, not assumed
    centre = ((full[0::2, 0::2].astype(int) + full[1::2, 0::2] + full[0::2, 1::2] + full[1::2, 1::2] + 2) // 4)
    assert psnr(centre, small) < 30.0
    assert psnr(full[1::2, 1::2], small) < 25.0


def test_resize_identity_and_edges(oracle):
    rng = np.random.default_rng(0)
    for dt in (np.uint8, np.uint16, np.float32):
        src = (rng.random((24, 36)) * 250).astype(dt)
        assert np.array_equal(oracle.resize_plane(src, 1, 36, 24), src)          # scale 1
        up = oracle.resize_plane(src, 1, 72, 48)
        assert np.array_equal(up[0::2, 0::2], src)                              # even samples exact
        c3 = oracle.resize_plane(src, 3, 4, 8)                                   # 12 px x3 -> 4 px
        assert c3.shape == (8, 12)
    nv = rng.integers(0, 256, (36, 48), dtype=np.uint8)                          # NV12 24x... host layout
    out = oracle.resize_surface(nv.reshape(-1), "NV12", 48, 24, 16, 8)
    assert out.size == 16 * 8 * 3 // 2
    assert np.array_equal(out[:128].reshape(8, 16), nv[:24][0::3, 0::3])
    uv = nv[24:].reshape(12, 24, 2)
    assert np.array_equal(out[128:].reshape(4, 8, 2), uv[0::3, 0::3])


# ---- Lanczos-3 restatement: properties that hold whatever NPP's exact taps are ----------------
def test_lanczos_weights_are_lanczos3():
    from oracle import oracle as o
    for a in np.linspace(0.0, 0.999, 41, dtype=np.float32):
        w = o.lanczos3_weights(float(a))
        t = np.float64(a) + np.array([2, 1, 0, -1, -2, -3.0])
        ref = np.sinc(t) * np.sinc(t / 3)
        ref /= ref.sum()
        assert np.abs(w - ref).max() < 3e-7
        assert abs(float(w.astype(np.float64).sum()) - 1.0) < 3e-7


def test_lanczos_identity_and_integer_factors():
    from oracle import oracle as o
    rng = np.random.default_rng(3)
    for dt, hi in ((np.uint8, 256), (np.uint16, 1024)):
        src = rng.integers(0, hi, (48, 66), dtype=dt)
        assert np.array_equal(o.resize_plane(src, 1, 66, 48, "lanczos"), src)              # a == 0 everywhere
        assert np.array_equal(o.resize_plane(src, 1, 33, 24, "lanczos"), src[::2, ::2])    # fixture geometry
        assert np.array_equal(o.resize_plane(src, 1, 22, 16, "lanczos"), src[::3, ::3])
        assert np.array_equal(o.resize_plane(src, 2, 33, 48, "lanczos"), o.resize_plane(src, 2, 33, 48, "linear"))
    flat = np.full((20, 30), 177, np.uint8)
    assert np.array_equal(o.resize_plane(flat, 1, 77, 41, "lanczos"), np.full((41, 77), 177, np.uint8))
    f32 = rng.random((16, 24), dtype=np.float32)
    assert np.array_equal(o.resize_plane(f32, 1, 24, 16, "lanczos"), f32)


def test_lanczos_upscale_is_sharper_than_bilinear_on_a_band_limited_signal():
    """sanity of the filter itself: reconstructing a smooth sinusoid, Lanczos-3 beats bilinear."""
    from oracle import oracle as o
    x = np.arange(64, dtype=np.float64)
    src = (127.5 + 100 * np.sin(2 * np.pi * x / 9.0))[None, :].repeat(8, 0).astype(np.float32)
    dw = 64 * 4
    xs = np.arange(dw) * (64 / dw)
    truth = 127.5 + 100 * np.sin(2 * np.pi * xs / 9.0)
    inner = slice(16, dw - 16)
    e_lz = np.abs(o.resize_plane(src, 1, dw, 8, "lanczos")[0][inner] - truth[inner]).max()
    e_bl = np.abs(o.resize_plane(src, 1, dw, 8, "linear")[0][inner] - truth[inner]).max()
    assert e_lz < 0.35 * e_bl
