"""PySurfaceRotator on the GPU vs the oracle (bit-exact), plus the reference's behaviour tests
(tests/test_PySurfaceRotator.py:63-94 unsupported params, :101-137 rotate vs etalons)."""
import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def upload(vali, gpu, fmt, w, h, host):
    s = vali.Surface.Make(fmt, w, h, gpu)
    assert vali.PyFrameUploader(gpu).Run(np.ascontiguousarray(host).reshape(-1).view(np.uint8), s)[0]
    return s


def download(vali, gpu, surf, dtype=np.uint8):
    out = np.zeros(surf.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(surf, out)[0]
    return out.view(dtype)


@pytest.mark.parametrize("angle", [90.0, 180.0, 270.0])
def test_rotate_reference_etalon(vali, gpu, oracle, angle):
    PIL = pytest.importorskip("PIL.Image")
    img = np.asarray(PIL.open(GOLDEN / "frame_0.jpg"))
    h, w, _ = img.shape
    src = upload(vali, gpu, vali.RGB, w, h, img)
    dw, dh = (w, h) if angle == 180.0 else (h, w)
    dst = vali.Surface.Make(vali.RGB, dw, dh, gpu)
    ok, info = vali.PySurfaceRotator(gpu).Run(src, dst, angle)
    assert ok and info == vali.TaskExecInfo.SUCCESS
    frame = download(vali, gpu, dst).reshape(dst.Shape)
    etalon = np.asarray(PIL.open(GOLDEN / f"frame_0_{int(angle)}_deg.jpg"))
    mse = np.mean((frame.astype(np.float64) - etalon) ** 2)
    assert 10 * np.log10(255.0 ** 2 / mse) >= 42.0
    assert np.array_equal(frame, np.rot90(img, k=int(angle) // 90))


@pytest.mark.parametrize("fmt,dtype", [("RGB", np.uint8), ("BGR", np.uint8), ("Y", np.uint8),
                                       ("RGB_32F", np.float32), ("YUV444", np.uint8),
                                       ("YUV420", np.uint8), ("YUV444_10bit", np.uint16)])
@pytest.mark.parametrize("angle", [0.0, 90.0, 180.0, 270.0, -90.0, 450.0])
@pytest.mark.parametrize("size", [(130, 70), (64, 64), (1920, 1080)])
def test_quarter_turns_bit_exact(vali, gpu, oracle, fmt, dtype, angle, size):
    w, h = size
    if size == (1920, 1080) and (fmt not in ("RGB", "YUV420") or angle not in (90.0, 270.0)):
        pytest.skip("full-size case only for the headline formats")
    pf = vali.PixelFormat[fmt]
    rng = np.random.default_rng(3)
    src = vali.Surface.Make(pf, w, h, gpu)
    host = (rng.random(src.HostSize // np.dtype(dtype).itemsize) * 255).astype(dtype)
    assert vali.PyFrameUploader(gpu).Run(host.view(np.uint8), src)[0]
    n = (int(round(angle)) + 360) % 360
    dw, dh = (w, h) if n in (0, 180) else (h, w)
    dst = vali.Surface.Make(pf, dw, dh, gpu)
    assert vali.PySurfaceRotator(gpu).Run(src, dst, angle) == (True, vali.TaskExecInfo.SUCCESS)
    got = download(vali, gpu, dst, dtype)
    off_s = off_d = 0
    ch = 3 if fmt in ("RGB", "BGR", "RGB_32F") else 1
    for sp, dp in zip(src.Planes, dst.Planes):
        pw, ph = sp.Width // ch, sp.Height
        plane = host[off_s: off_s + sp.Width * sp.Height].reshape(ph, sp.Width)
        a, sx, sy = oracle.canonical_shifts(angle, pw, ph)
        want = oracle.rotate_plane(np.ascontiguousarray(plane), ch, dp.Width // ch, dp.Height, a, sx, sy)
        assert np.array_equal(got[off_d: off_d + dp.Width * dp.Height].reshape(dp.Height, dp.Width), want)
        off_s += sp.Width * sp.Height
        off_d += dp.Width * dp.Height


@pytest.mark.parametrize("angle,shift", [(30.0, (40.0, 10.0)), (45.5, (0.0, 60.0)), (-12.25, (3.5, 7.25)),
                                         (90.0, (5.0, 100.0)), (180.0, (50.0, 50.0))])
@pytest.mark.parametrize("fmt,dtype,ch", [("RGB", np.uint8, 3), ("Y", np.uint8, 1), ("RGB_32F", np.float32, 3)])
def test_arbitrary_angles_bit_exact(vali, gpu, oracle, angle, shift, fmt, dtype, ch):
    w, h, dw, dh = 150, 90, 170, 140
    pf = vali.PixelFormat[fmt]
    rng = np.random.default_rng(4)
    host = (rng.random((h, w * ch)) * 255).astype(dtype)
    src = upload(vali, gpu, pf, w, h, host)
    dst = vali.Surface.Make(pf, dw, dh, gpu)
    fill = np.full(dst.HostSize // np.dtype(dtype).itemsize, 77, dtype)
    assert vali.PyFrameUploader(gpu).Run(fill.view(np.uint8), dst)[0]
    ok, info = vali.PySurfaceRotator(gpu).Run(src, dst, angle, shift[0], shift[1])
    assert ok
    got = download(vali, gpu, dst, dtype).reshape(dh, dw * ch)
    want = oracle.rotate_plane(host, ch, dw, dh, angle, shift[0], shift[1], fill=77)
    assert np.array_equal(got, want)


def test_unsupported_params(vali, gpu):
    rot = vali.PySurfaceRotator(gpu)
    a = vali.Surface.Make(vali.NV12, 64, 48, gpu)
    b = vali.Surface.Make(vali.NV12, 48, 64, gpu)
    assert rot.Run(src=a, dst=b, angle=90.0) == (False, vali.TaskExecInfo.NOT_SUPPORTED)
    c = vali.Surface.Make(vali.RGB, 48, 64, gpu)
    assert rot.Run(a, c, 90.0) == (False, vali.TaskExecInfo.SRC_DST_FMT_MISMATCH)
    p = vali.Surface.Make(vali.RGB_PLANAR, 64, 48, gpu)
    q = vali.Surface.Make(vali.RGB_PLANAR, 48, 64, gpu)
    assert rot.Run(p, q, 90.0) == (False, vali.TaskExecInfo.INVALID_INPUT)   # RotateSurface.cpp:135-136
    assert vali.PixelFormat.RGB in rot.SupportedFormats and vali.PixelFormat.NV12 not in rot.SupportedFormats


@pytest.mark.parametrize("angle", [90.0, 270.0])
@pytest.mark.parametrize("size", [(2432, 1792), (2435, 1731)])
@pytest.mark.parametrize("fmt", ["RGB", "BGR"])
def test_quarter_turns_of_large_packed_frames(vali, gpu, fmt, angle, size):
    """Frames of 4 Mpixel and more take the 64x128-pixel tile form of the 3-byte transpose (rotate.hip kRotTallPixels): a
    whole-tile and a ragged geometry, single call and batch, against numpy's rot90 (a quarter turn is a permutation)."""
    w, h = size
    pf = vali.PixelFormat[fmt]
    rot = vali.PySurfaceRotator(gpu)
    srcs = [vali.Surface.Make(pf, w, h, gpu) for _ in range(2)]
    dsts = [vali.Surface.Make(pf, h, w, gpu) for _ in range(3)]
    imgs = []
    for i, s in enumerate(srcs):
        host = np.random.default_rng(40 + i).integers(0, 256, s.HostSize, dtype=np.uint8)
        imgs.append(host.reshape(h, w, 3))
        assert vali.PyFrameUploader(gpu).Run(host, s)[0]
    assert rot.Run(srcs[0], dsts[0], angle) == (True, vali.TaskExecInfo.SUCCESS)
    assert rot.RunBatch(srcs, dsts[1:], angle) == (True, vali.TaskExecInfo.SUCCESS)
    k = 1 if angle == 90.0 else 3
    # the reference's 90 degree turn = NPP's rotation about the origin + shift: dst(x', y') = src(W-1-y', x') = rot90 k=1
    for d, img in zip(dsts, [imgs[0], imgs[0], imgs[1]]):
        got = download(vali, gpu, d).reshape(w, h, 3)
        assert np.array_equal(got, np.rot90(img, k=k))


@pytest.mark.parametrize("fmt,angle", [("RGB", 90.0), ("RGB", 270.0), ("YUV420", 90.0), ("RGB", 33.0)])
def test_rotate_batch(vali, gpu, oracle, fmt, angle):
    """One launch over a batch == per-surface Run."""
    w, h, n = 320, 180, 5
    pf = vali.PixelFormat[fmt]
    rot = vali.PySurfaceRotator(gpu)
    quarter = angle in (90.0, 270.0)
    dw, dh = (h, w) if quarter else (w, h)
    srcs = [vali.Surface.Make(pf, w, h, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(pf, dw, dh, gpu) for _ in range(n)]
    refs = [vali.Surface.Make(pf, dw, dh, gpu) for _ in range(n)]
    for i, s in enumerate(srcs):
        host = np.random.default_rng(i).integers(0, 256, s.HostSize, dtype=np.uint8)
        assert vali.PyFrameUploader(gpu).Run(host, s)[0]
        for t in (dsts[i], refs[i]):
            assert vali.PyFrameUploader(gpu).Run(np.full(t.HostSize, 9, np.uint8), t)[0]
    sx, sy = (0.0, 0.0) if quarter else (20.0, 50.0)
    assert rot.RunBatch(srcs, dsts, angle, sx, sy) == (True, vali.TaskExecInfo.SUCCESS)
    for s, d, r in zip(srcs, dsts, refs):
        assert rot.Run(s, r, angle, sx, sy)[0]
        assert np.array_equal(download(vali, gpu, d), download(vali, gpu, r))
