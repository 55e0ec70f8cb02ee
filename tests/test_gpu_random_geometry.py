"""Randomised geometry sweep (seeded): sizes, scale factors and tile-boundary cases nobody wrote
down by hand, every result bit-exact against the oracle.  Catches disagreements in the
float coordinate math (host span prediction vs device, staged vs gather paths, ragged tiles)."""
import numpy as np
import pytest

from conftest import make_nv12

pytestmark = pytest.mark.gpu


def _cases(seed, n, lo=2, hi=700, even=True):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        v = rng.integers(lo, hi, 4)
        if rng.random() < 0.3:                      # tile-boundary widths: k*256 + {-2..2}
            v[2] = max(lo, int(rng.integers(1, 4)) * 256 + int(rng.integers(-2, 3)))
        if rng.random() < 0.2:                      # extreme aspect
            v[1] = int(rng.integers(lo, 12))
        if even:
            v = (v // 2) * 2
            v = np.maximum(v, 2)
        out.append(tuple(int(x) for x in v))
    return out


def _up(vali, gpu, fmt, w, h, host):
    s = vali.Surface.Make(fmt, w, h, gpu)
    assert vali.PyFrameUploader(gpu).Run(np.ascontiguousarray(host).reshape(-1).view(np.uint8), s)[0]
    return s


def _down(vali, gpu, s, dt=np.uint8):
    out = np.zeros(s.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(s, out)[0]
    return out.view(dt)


@pytest.mark.parametrize("geom", _cases(101, 120))
def test_random_nv12_resize_ud_preproc(vali, gpu, oracle, geom):
    sw, sh, dw, dh = geom
    nv = make_nv12(sw, sh, sw * 31 + dh)
    src = _up(vali, gpu, vali.NV12, sw, sh, nv)
    flat = nv.reshape(-1)
    for interp, name in ((vali.Interpolation.LINEAR, "linear"), (vali.Interpolation.CUBIC, "cubic"),
                         (vali.Interpolation.LANCZOS, "lanczos")):
        d = vali.Surface.Make(vali.NV12, dw, dh, gpu)
        assert vali.PySurfaceResizer(vali.NV12, gpu, interpolation=interp).Run(src, d)[0]
        assert np.array_equal(_down(vali, gpu, d), oracle.resize_surface(flat, "NV12", sw, sh, dw, dh, name)), name
    ud = vali.PySurfaceUD(gpu)
    for fmt, dt in (("RGB", np.uint8), ("RGB_32F_PLANAR", np.float32), ("YUV444", np.uint8)):
        d = vali.Surface.Make(vali.PixelFormat[fmt], dw, dh, gpu)
        assert ud.Run(src, d)[0]
        want = oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, fmt).reshape(-1)
        assert np.array_equal(_down(vali, gpu, d, dt).view(np.uint8), want.view(np.uint8)), fmt
    rgb = oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, "RGB").reshape(dh, dw, 3)
    for k, angle in ((1, 90.0), (2, 180.0), (3, 270.0)):
        rw, rh = (dh, dw) if k & 1 else (dw, dh)
        d = vali.Surface.Make(vali.RGB, rw, rh, gpu)
        assert ud.RunRotated(src, d, angle)[0]
        assert np.array_equal(_down(vali, gpu, d).reshape(rh, rw, 3), np.rot90(rgb, k=k)), angle
    from vali_amd.tasks import CSC_NPP_709HDTV
    f = vali.Surface.Make(vali.RGB_32F_PLANAR, dw, dh, gpu)
    assert vali.PySurfacePreprocessor(gpu, mean=(0.5, 0.4, 0.3), std=(0.2, 0.3, 0.4), div=2.0).Run(src, f)[0]
    n12 = nv if (dw, dh) == (sw, sh) else oracle.resize_surface(flat, "NV12", sw, sh, dw, dh).reshape(dh * 3 // 2, dw)
    q = oracle.nv12_to_rgb(n12, dw, dh, oracle.csc_from_tuple(CSC_NPP_709HDTV), "RGB").reshape(dh, dw, 3)
    x = (q.astype(np.float32) / np.float32(255)).transpose(2, 0, 1) / np.float32(2.0)
    want = (x - np.float32([0.5, 0.4, 0.3])[:, None, None]) / np.float32([0.2, 0.3, 0.4])[:, None, None]
    assert np.array_equal(_down(vali, gpu, f, np.float32).view(np.uint32), want.astype(np.float32).reshape(-1).view(np.uint32))


@pytest.mark.parametrize("geom", _cases(202, 60, lo=1, hi=500, even=False))
@pytest.mark.parametrize("fmt,dt,ch", [("RGB", np.uint8, 3), ("Y", np.uint8, 1), ("RGB_32F", np.float32, 3)])
def test_random_single_plane_resize_rotate(vali, gpu, oracle, geom, fmt, dt, ch):
    sw, sh, dw, dh = geom
    rng = np.random.default_rng(sw * 13 + dw)
    host = (rng.random(sw * sh * ch) * 255).astype(dt)
    pf = vali.PixelFormat[fmt]
    src = _up(vali, gpu, pf, sw, sh, host)
    for interp, name in ((vali.Interpolation.LINEAR, "linear"), (vali.Interpolation.CUBIC, "cubic"),
                         (vali.Interpolation.LANCZOS, "lanczos")):
        d = vali.Surface.Make(pf, dw, dh, gpu)
        assert vali.PySurfaceResizer(pf, gpu, interpolation=interp).Run(src, d)[0]
        want = oracle.resize_surface(host, fmt, sw, sh, dw, dh, name)
        assert np.array_equal(_down(vali, gpu, d, dt).view(np.uint8), want.view(np.uint8)), name
    img = host.reshape(sh, sw, ch)
    for k, angle in ((1, 90.0), (2, 180.0), (3, 270.0)):
        rw, rh = (sh, sw) if k & 1 else (sw, sh)
        d = vali.Surface.Make(pf, rw, rh, gpu)
        assert vali.PySurfaceRotator(gpu).Run(src, d, angle)[0]
        assert np.array_equal(_down(vali, gpu, d, dt).reshape(rh, rw, ch).view(np.uint8), np.rot90(img, k=k).copy().view(np.uint8)), angle


def test_no_resize_of_this_session_ran_without_its_tap_table(vali, gpu):
    """Runs after the sweeps above (300 geometries x 2 filters x up to 4 axes -- more tables than round 5's cache of 256 could hold):
    every Lanczos / bicubic call outside a graph capture got its tap table -- old ones are evicted, none is refused (VERDICT r05 #6;
    vali_amd/csrc/tap_table.hip, VALI_TUNE_TAP_FALLBACKS)."""
    assert vali.tuning.Get("TAP_FALLBACKS") == 0, (vali.tuning.Get("TAP_FALLBACKS"), vali.tuning.Get("TAP_EVICTIONS"))
