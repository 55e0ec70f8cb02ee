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


# ---- bicubic restatement (Keys a = -1/2): pinned against an independent float64 evaluation ----
def _keys(t, a=-0.5):
    t = np.abs(t)
    return np.where(t <= 1, (a + 2) * t ** 3 - (a + 3) * t ** 2 + 1,
                    np.where(t < 2, a * t ** 3 - 5 * a * t ** 2 + 8 * a * t - 4 * a, 0.0))


def test_cubic_weights_are_the_keys_kernel():
    from oracle import oracle as o
    for a in np.linspace(0.0, 0.999, 41, dtype=np.float32):
        w = o.cubic_weights(float(a))
        ref = _keys(np.float64(a) + np.array([1, 0, -1, -2.0]))
        assert np.abs(w - ref).max() < 2e-7
        assert abs(float(w.astype(np.float64).sum()) - 1.0) < 3e-7
    assert np.array_equal(o.cubic_weights(0.0), np.array([0, 1, 0, 0], np.float32))


def test_cubic_identity_integer_factors_and_float64_model():
    from oracle import oracle as o
    rng = np.random.default_rng(4)
    for dt, hi in ((np.uint8, 256), (np.uint16, 1024)):
        src = rng.integers(0, hi, (48, 66), dtype=dt)
        assert np.array_equal(o.resize_plane(src, 1, 66, 48, "cubic"), src)
        assert np.array_equal(o.resize_plane(src, 1, 33, 24, "cubic"), src[::2, ::2])
        assert np.array_equal(o.resize_plane(src, 2, 33, 48, "cubic"), o.resize_plane(src, 2, 33, 48, "linear"))
    flat = np.full((20, 30), 177, np.uint8)
    assert np.array_equal(o.resize_plane(flat, 1, 77, 41, "cubic"), np.full((41, 77), 177, np.uint8))
    # separable float64 model on the same grid with clamped indices
    src = rng.random((23, 31), dtype=np.float32)
    dw, dh = 50, 37

    def taps(n_dst, n_src):
        f = np.arange(n_dst, dtype=np.float32) * (np.float32(n_src) / np.float32(n_dst))
        i = np.floor(f).astype(int)
        a = (f - np.floor(f)).astype(np.float64)
        idx = np.clip(i[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
        return idx, _keys(a[:, None] - np.arange(-1, 3)[None, :])

    ix, wx = taps(dw, 31)
    iy, wy = taps(dh, 23)
    hor = (src.astype(np.float64)[:, ix] * wx[None]).sum(-1)          # (23, dw)
    model = (hor[iy] * wy[:, :, None]).sum(1)                          # (dh, dw)
    assert np.abs(o.resize_plane(src, 1, dw, dh, "cubic") - model).max() < 2e-6


def test_cubic_sits_between_bilinear_and_lanczos_on_a_band_limited_signal():
    from oracle import oracle as o
    x = np.arange(64, dtype=np.float64)
    src = (127.5 + 100 * np.sin(2 * np.pi * x / 9.0))[None, :].repeat(8, 0).astype(np.float32)
    dw = 64 * 4
    xs = np.arange(dw) * (64 / dw)
    truth = 127.5 + 100 * np.sin(2 * np.pi * xs / 9.0)
    inner = slice(16, dw - 16)
    err = {k: np.abs(o.resize_plane(src, 1, dw, 8, k)[0][inner] - truth[inner]).max() for k in ("linear", "cubic", "lanczos")}
    assert err["lanczos"] < err["cubic"] < err["linear"]
