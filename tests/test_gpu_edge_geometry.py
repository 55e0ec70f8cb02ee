"""Edge geometry for the gather kernels (UD, bilinear / Lanczos resize, fused pre-processing):
foreign memory with unaligned rows (the kernels' byte / direct-gather fallbacks), very wide and
very small surfaces, scale factors beyond the LDS strip.  All bit-exact against the oracle.
(The reference tests only sane video sizes; these are the shapes its NPP calls would accept.)"""
import numpy as np
import pytest

from conftest import make_nv12

pytestmark = pytest.mark.gpu


def foreign_nv12(vali, w, h, nv12, pad=3, skew=0):
    """NV12 surface borrowed from a torch tensor whose rows are `w + pad` bytes apart and whose
    base is `skew` bytes off alignment."""
    import torch

    rows = h * 3 // 2
    raw = torch.zeros(rows * (w + pad) + 64, dtype=torch.uint8, device="cuda")
    view = raw[skew: skew + rows * (w + pad)].view(rows, w + pad)[:, :w]
    view.copy_(torch.from_numpy(np.ascontiguousarray(nv12)))
    torch.cuda.synchronize()
    s = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(view), vali.NV12)
    assert s.Pitch == w + pad and not s.IsOwnMemory
    return s, raw


def download(vali, gpu, surf, dtype=np.uint8):
    out = np.zeros(surf.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(surf, out)[0]
    return out.view(dtype)


def upload(vali, gpu, fmt, w, h, host):
    s = vali.Surface.Make(fmt, w, h, gpu)
    assert vali.PyFrameUploader(gpu).Run(np.ascontiguousarray(host).reshape(-1).view(np.uint8), s)[0]
    return s


@pytest.mark.parametrize("pad,skew", [(3, 0), (16, 5), (1, 7)])
def test_foreign_unaligned_rows_all_gather_kernels(vali, gpu, oracle, pad, skew):
    sw, sh, dw, dh = 320, 180, 214, 120
    nv = make_nv12(sw, sh, 5)
    src, keep = foreign_nv12(vali, sw, sh, nv, pad, skew)
    flat = nv.reshape(-1)
    # UD
    dst = vali.Surface.Make(vali.RGB, dw, dh, gpu)
    assert vali.PySurfaceUD(gpu).Run(src, dst)[0]
    assert np.array_equal(download(vali, gpu, dst), oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, "RGB").reshape(-1))
    # resize, both filters
    for interp, name in ((vali.Interpolation.LINEAR, "linear"), (vali.Interpolation.CUBIC, "cubic"),
                         (vali.Interpolation.LANCZOS, "lanczos")):
        small = vali.Surface.Make(vali.NV12, dw, dh, gpu)
        assert vali.PySurfaceResizer(vali.NV12, gpu, interpolation=interp).Run(src, small)[0]
        assert np.array_equal(download(vali, gpu, small), oracle.resize_surface(flat, "NV12", sw, sh, dw, dh, name))
    # fused pre-processing, same size and resized
    from vali_amd.tasks import CSC_NPP_709HDTV
    for (ow, oh) in ((sw, sh), (dw, dh)):
        f = vali.Surface.Make(vali.RGB_32F_PLANAR, ow, oh, gpu)
        assert vali.PySurfacePreprocessor(gpu, mean=(0.1, 0.2, 0.3), std=(0.5, 0.25, 2.0), div=1.0).Run(src, f)[0]
        n12 = nv if (ow, oh) == (sw, sh) else oracle.resize_surface(flat, "NV12", sw, sh, ow, oh).reshape(oh * 3 // 2, ow)
        rgb = oracle.nv12_to_rgb(n12, ow, oh, oracle.csc_from_tuple(CSC_NPP_709HDTV), "RGB").reshape(oh, ow, 3)
        x = (rgb.astype(np.float32) / np.float32(255)).transpose(2, 0, 1) / np.float32(1.0)
        want = (x - np.float32([0.1, 0.2, 0.3])[:, None, None]) / np.float32([0.5, 0.25, 2.0])[:, None, None]
        assert np.array_equal(download(vali, gpu, f, np.float32).view(np.uint32), want.astype(np.float32).reshape(-1).view(np.uint32))
    del keep


@pytest.mark.parametrize("geom", [(8192, 64, 2730, 22), (8192, 64, 8192, 64), (4096, 256, 128, 16),
                                  (4096, 256, 126, 10), (2, 2, 2, 2), (2, 2, 34, 18), (6, 4, 2, 2),
                                  (258, 18, 254, 34)])
def test_extreme_sizes_nv12(vali, gpu, oracle, geom):
    sw, sh, dw, dh = geom
    nv = make_nv12(sw, sh, 17)
    src = upload(vali, gpu, vali.NV12, sw, sh, nv)
    flat = nv.reshape(-1)
    for interp, name in ((vali.Interpolation.LINEAR, "linear"), (vali.Interpolation.CUBIC, "cubic"),
                         (vali.Interpolation.LANCZOS, "lanczos")):
        small = vali.Surface.Make(vali.NV12, dw, dh, gpu)
        assert vali.PySurfaceResizer(vali.NV12, gpu, interpolation=interp).Run(src, small)[0]
        assert np.array_equal(download(vali, gpu, small), oracle.resize_surface(flat, "NV12", sw, sh, dw, dh, name)), name
    for dst_fmt in ("RGB", "YUV444", "RGB_32F_PLANAR"):
        d = vali.Surface.Make(vali.PixelFormat[dst_fmt], dw, dh, gpu)
        assert vali.PySurfaceUD(gpu).Run(src, d)[0]
        dt = np.float32 if "32F" in dst_fmt else np.uint8
        got = download(vali, gpu, d, dt)
        want = oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, dst_fmt).reshape(-1)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), dst_fmt
    from vali_amd.tasks import CSC_NPP_709HDTV
    f = vali.Surface.Make(vali.RGB_32F, dw, dh, gpu)
    assert vali.PySurfacePreprocessor(gpu).Run(src, f)[0]
    n12 = nv if (dw, dh) == (sw, sh) else oracle.resize_surface(flat, "NV12", sw, sh, dw, dh).reshape(dh * 3 // 2, dw)
    rgb = oracle.nv12_to_rgb(n12, dw, dh, oracle.csc_from_tuple(CSC_NPP_709HDTV), "RGB").reshape(-1)
    want = rgb.astype(np.float32) / np.float32(255)
    assert np.array_equal(download(vali, gpu, f, np.float32).view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("fmt,dt,ch", [("Y", np.uint8, 1), ("RGB", np.uint8, 3), ("RGB_32F", np.float32, 3)])
@pytest.mark.parametrize("geom", [(1, 1, 5, 3), (3, 2, 1, 1), (5000, 3, 313, 7), (17, 9, 600, 2)])
def test_extreme_sizes_single_plane(vali, gpu, oracle, fmt, dt, ch, geom):
    sw, sh, dw, dh = geom
    rng = np.random.default_rng(sw * 7 + dh)
    host = (rng.random(sw * sh * ch) * 255).astype(dt)
    src = upload(vali, gpu, vali.PixelFormat[fmt], sw, sh, host)
    for interp, name in ((vali.Interpolation.LINEAR, "linear"), (vali.Interpolation.CUBIC, "cubic"),
                         (vali.Interpolation.LANCZOS, "lanczos")):
        d = vali.Surface.Make(vali.PixelFormat[fmt], dw, dh, gpu)
        assert vali.PySurfaceResizer(vali.PixelFormat[fmt], gpu, interpolation=interp).Run(src, d)[0]
        want = oracle.resize_surface(host, fmt, sw, sh, dw, dh, name)
        assert np.array_equal(download(vali, gpu, d, dt).view(np.uint8), want.view(np.uint8)), name


def test_batches_mix_pitches_and_alignment(vali, gpu, oracle):
    """A batch is an array of per-surface descriptors: surfaces of one batch may differ in pitch and
    alignment (library-allocated next to foreign unaligned memory) and each keeps its own path."""
    sw, sh, dw, dh = 320, 180, 160, 90
    nvs = [make_nv12(sw, sh, 60 + i) for i in range(4)]
    keep, srcs = [], []
    for i, nv in enumerate(nvs):
        if i % 2:
            s, raw = foreign_nv12(vali, sw, sh, nv, pad=5 + i, skew=3)
            keep.append(raw)
        else:
            s = upload(vali, gpu, vali.NV12, sw, sh, nv)
        srcs.append(s)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    rgb = [vali.Surface.Make(vali.RGB, sw, sh, gpu) for _ in nvs]
    assert vali.PySurfaceConverter(gpu).RunBatch(srcs, rgb, cc)[0]
    small = [vali.Surface.Make(vali.NV12, dw, dh, gpu) for _ in nvs]
    assert vali.PySurfaceResizer(vali.NV12, gpu, interpolation=vali.Interpolation.LANCZOS).RunBatch(srcs, small)[0]
    ud = [vali.Surface.Make(vali.RGB_PLANAR, dw, dh, gpu) for _ in nvs]
    assert vali.PySurfaceUD(gpu).RunBatch(srcs, ud)[0]
    turned = [vali.Surface.Make(vali.RGB, dh, dw, gpu) for _ in nvs]
    assert vali.PySurfaceUD(gpu).RunRotatedBatch(srcs, turned, angle=90.0)[0]
    pre = [vali.Surface.Make(vali.RGB_PLANAR, dw, dh, gpu) for _ in nvs]
    assert vali.PySurfacePreprocessor(gpu).RunBatch(srcs, pre, cc)[0]
    for i, nv in enumerate(nvs):
        flat = nv.reshape(-1)
        assert np.array_equal(download(vali, gpu, rgb[i]), oracle.nv12_to_rgb(nv, sw, sh, oracle.csc(1), "RGB").reshape(-1))
        assert np.array_equal(download(vali, gpu, small[i]), oracle.resize_surface(flat, "NV12", sw, sh, dw, dh, "lanczos"))
        assert np.array_equal(download(vali, gpu, ud[i]), oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, "RGB_PLANAR").reshape(-1))
        want = np.rot90(oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, "RGB").reshape(dh, dw, 3), k=1)
        assert np.array_equal(download(vali, gpu, turned[i]).reshape(dw, dh, 3), want)
        n12 = oracle.resize_surface(flat, "NV12", sw, sh, dw, dh).reshape(dh * 3 // 2, dw)
        assert np.array_equal(download(vali, gpu, pre[i]), oracle.nv12_to_rgb(n12, dw, dh, oracle.csc(1), "RGB_PLANAR").reshape(-1))
    del keep


def test_odd_sizes_of_subsampled_formats_are_refused_everywhere(vali, gpu):
    """ONE rule (include/vali_hip.h): 4:2:0 surfaces need an even width and height, YUV422 an even width -- converter,
    resizer, UD (single and batch), fused pre-processing all answer (False, INVALID_INPUT); Surface.Make itself allocates any
    size like the reference's (the entry points used to disagree)."""
    T = vali.TaskExecInfo
    odd = vali.Surface.Make(vali.NV12, 65, 49, gpu)
    even = vali.Surface.Make(vali.NV12, 64, 48, gpu)
    rgb_odd, rgb = vali.Surface.Make(vali.RGB, 65, 49, gpu), vali.Surface.Make(vali.RGB, 64, 48, gpu)
    assert vali.PySurfaceConverter(gpu).Run(odd, rgb_odd, None) == (False, T.INVALID_INPUT)
    for interp in (vali.Interpolation.LINEAR, vali.Interpolation.LANCZOS):
        rs = vali.PySurfaceResizer(vali.NV12, gpu, interpolation=interp)
        assert rs.Run(odd, even) == (False, T.INVALID_INPUT)
        assert rs.Run(even, odd) == (False, T.INVALID_INPUT)
        assert rs.RunBatch([odd, odd], [even, even]) == (False, T.INVALID_INPUT)
        assert rs.Run(even, vali.Surface.Make(vali.NV12, 32, 24, gpu)) == (True, T.SUCCESS)
    ud = vali.PySurfaceUD(gpu)
    assert ud.Run(odd, rgb) == (False, T.INVALID_INPUT)
    assert ud.RunBatch([odd], [rgb]) == (False, T.INVALID_INPUT)
    assert ud.RunRotatedBatch([odd], [rgb], 180.0) == (False, T.INVALID_INPUT)
    assert ud.Run(even, rgb_odd) == (True, T.SUCCESS)          # the 4:4:4 side may have any size
    y422 = vali.Surface.Make(vali.YUV422, 33, 20, gpu)
    assert vali.PySurfaceResizer(vali.YUV422, gpu).Run(y422, vali.Surface.Make(vali.YUV422, 32, 21, gpu)) == (False, T.INVALID_INPUT)
    assert vali.PySurfaceResizer(vali.YUV422, gpu).Run(vali.Surface.Make(vali.YUV422, 34, 21, gpu),
                                                       vali.Surface.Make(vali.YUV422, 32, 19, gpu)) == (True, T.SUCCESS)
