"""PySurfacePreprocessor (fused NV12 -> resize -> RGB -> float -> normalise) on the GPU.

The fused task is DEFINED as the chain the reference's samples run
(reference tests/test_TorchSegmentation.py:176-240: PySurfaceConverter NV12->RGB -> RGB_32F
-> RGB_32F_PLANAR, torch.divide(x, 255.0), torchvision Normalize), optionally behind a
PySurfaceResizer.  Checked bit-exactly (float32) against
  (a) that chain run with this library's own GPU tasks + float32 numpy for the torch part,
  (b) the CPU oracle composition (oracle resize -> oracle nv12_to_rgb -> float32 numpy)."""
import numpy as np
import pytest

from conftest import make_nv12

pytestmark = pytest.mark.gpu

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)   # the reference test's constants


def torch_part(planar_u8_over_255: np.ndarray, div, mean, std) -> np.ndarray:
    """torch.divide(x, div) ; (x - mean[c]) / std[c] in float32 (IEEE, like torch on CPU/GPU)."""
    x = planar_u8_over_255.astype(np.float32) / np.float32(div)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((x - m) / s).astype(np.float32)


def upload(vali, gpu, host, w, h):
    s = vali.Surface.Make(vali.NV12, w, h, gpu)
    assert vali.PyFrameUploader(gpu).Run(np.ascontiguousarray(host).reshape(-1), s)[0]
    return s


def download(vali, gpu, surf, dtype):
    out = np.zeros(surf.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(surf, out)[0]
    return out.view(dtype)


def chain_gpu(vali, gpu, src, dw, dh, cc, div, mean, std):
    sw, sh = src.Width, src.Height
    cur = src
    if (sw, sh) != (dw, dh):
        small = vali.Surface.Make(vali.NV12, dw, dh, gpu)
        assert vali.PySurfaceResizer(vali.NV12, gpu, interpolation=vali.Interpolation.LINEAR).Run(cur, small)[0]
        cur = small
    cvt = vali.PySurfaceConverter(gpu)
    rgb = vali.Surface.Make(vali.RGB, dw, dh, gpu)
    f32 = vali.Surface.Make(vali.RGB_32F, dw, dh, gpu)
    pl = vali.Surface.Make(vali.RGB_32F_PLANAR, dw, dh, gpu)
    assert cvt.Run(cur, rgb, cc)[0] and cvt.Run(rgb, f32)[0] and cvt.Run(f32, pl)[0]
    x = download(vali, gpu, pl, np.float32).reshape(3, dh, dw)
    return torch_part(x, div, mean, std)


def chain_oracle(oracle, host, sw, sh, dw, dh, coeffs, div, mean, std):
    nv12 = host if (sw, sh) == (dw, dh) else oracle.resize_surface(
        np.ascontiguousarray(host).reshape(-1), "NV12", sw, sh, dw, dh).reshape(dh * 3 // 2, dw)
    rgb = oracle.nv12_to_rgb(nv12, dw, dh, oracle.csc_from_tuple(coeffs), "RGB").reshape(dh, dw, 3)
    x = (rgb.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)
    return torch_part(x, div, mean, std)


GEOMS = [(640, 360, 640, 360), (1920, 1080, 1920, 1080), (1920, 1080, 640, 384), (848, 464, 300, 300),
         (130, 70, 58, 34), (640, 360, 1000, 500), (3840, 2160, 1280, 720), (66, 34, 66, 34),
         # the resized path fetches both horizontal taps with one load: narrow sources, the clamped last
         # column under strong upscaling, odd tap positions
         (2, 2, 8, 6), (4, 4, 6, 10), (6, 2, 2, 2), (8, 4, 258, 130), (34, 18, 1030, 70), (3840, 2160, 640, 640)]


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("norm", [(255.0, MEAN, STD), (1.0, (0, 0, 0), (1, 1, 1))])
def test_preproc_equals_chain(vali, gpu, oracle, geom, norm):
    sw, sh, dw, dh = geom
    div, mean, std = norm
    host = make_nv12(sw, sh, seed=geom[0] + geom[3])
    src = upload(vali, gpu, host, sw, sh)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    pp = vali.PySurfacePreprocessor(gpu, mean=mean, std=std, div=div)
    dst = vali.Surface.Make(vali.RGB_32F_PLANAR, dw, dh, gpu)
    assert pp.Run(src, dst, cc) == (True, vali.TaskExecInfo.SUCCESS)
    got = download(vali, gpu, dst, np.float32).reshape(3, dh, dw)
    want_gpu = chain_gpu(vali, gpu, src, dw, dh, cc, div, mean, std)
    assert np.array_equal(got.view(np.uint32), want_gpu.view(np.uint32))
    from vali_amd.tasks import CSC_NPP_709CSC
    want_cpu = chain_oracle(oracle, host, sw, sh, dw, dh, CSC_NPP_709CSC, div, mean, std)
    assert np.array_equal(got.view(np.uint32), want_cpu.view(np.uint32))


@pytest.mark.parametrize("geom", [(640, 360, 640, 360), (1920, 1080, 512, 288)])
def test_preproc_packed_and_default_ctx(vali, gpu, oracle, geom):
    sw, sh, dw, dh = geom
    host = make_nv12(sw, sh, seed=9)
    src = upload(vali, gpu, host, sw, sh)
    pp = vali.PySurfacePreprocessor(gpu, mean=MEAN, std=STD, div=255.0)
    dst = vali.Surface.Make(vali.RGB_32F, dw, dh, gpu)
    assert pp.RunAsync(src, dst) == (True, vali.TaskExecInfo.SUCCESS)     # default: BT.709 full range
    ev = vali.CudaStreamEvent(pp.Stream, gpu)
    ev.Record()
    ev.Wait()
    got = download(vali, gpu, dst, np.float32).reshape(dh, dw, 3).transpose(2, 0, 1)
    from vali_amd.tasks import CSC_NPP_709HDTV
    want = chain_oracle(oracle, host, sw, sh, dw, dh, CSC_NPP_709HDTV, 255.0, MEAN, STD)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_preproc_batch(vali, gpu, oracle):
    sw, sh, dw, dh, n = 1920, 1080, 640, 640, 5
    hosts = [make_nv12(sw, sh, seed=40 + i) for i in range(2)]
    srcs = [upload(vali, gpu, hosts[i % 2], sw, sh) for i in range(n)]
    dsts = [vali.Surface.Make(vali.RGB_32F_PLANAR, dw, dh, gpu) for _ in range(n)]
    pp = vali.PySurfacePreprocessor(gpu, mean=MEAN, std=STD, div=255.0)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_601, vali.ColorRange.JPEG)
    assert pp.RunBatch(srcs, dsts, cc) == (True, vali.TaskExecInfo.SUCCESS)
    from vali_amd.tasks import CSC_NPP_YUV
    wants = [chain_oracle(oracle, h, sw, sh, dw, dh, CSC_NPP_YUV, 255.0, MEAN, STD) for h in hosts]
    for i, d in enumerate(dsts):
        got = download(vali, gpu, d, np.float32).reshape(3, dh, dw)
        assert np.array_equal(got.view(np.uint32), wants[i % 2].view(np.uint32))


def test_preproc_errors(vali, gpu):
    pp = vali.PySurfacePreprocessor(gpu)
    nv12 = vali.Surface.Make(vali.NV12, 64, 48, gpu)
    assert pp.Run(nv12, vali.Surface.Make(vali.YUV444, 64, 48, gpu)) == (False, vali.TaskExecInfo.NOT_SUPPORTED)
    assert pp.Run(vali.Surface.Make(vali.YUV420, 64, 48, gpu),
                  vali.Surface.Make(vali.RGB_32F_PLANAR, 64, 48, gpu)) == (False, vali.TaskExecInfo.NOT_SUPPORTED)
    assert pp.Run(nv12, vali.Surface.Make(vali.RGB_32F_PLANAR, 31, 24, gpu)) == (False, vali.TaskExecInfo.INVALID_INPUT)
    bad = vali.ColorspaceConversionContext(vali.ColorSpace.BT_601, vali.ColorRange.MPEG)
    assert pp.Run(nv12, vali.Surface.Make(vali.RGB_32F_PLANAR, 64, 48, gpu), bad) == \
        (False, vali.TaskExecInfo.UNSUPPORTED_FMT_CONV_PARAMS)                # nv12_rgb's rule, :144-148
    with pytest.raises(ValueError):
        vali.PySurfacePreprocessor(gpu, std=(1.0, 0.0, 1.0))


@pytest.mark.parametrize("dst_fmt", ["RGB", "BGR", "RGB_PLANAR"])
@pytest.mark.parametrize("geom", [(1920, 1080, 1920, 1080), (1920, 1080, 640, 384), (848, 464, 300, 300),
                                  (130, 70, 58, 34), (640, 360, 1000, 500), (3840, 2160, 1280, 720)])
def test_preproc_u8_equals_resizer_plus_converter(vali, gpu, oracle, dst_fmt, geom):
    """8-bit destinations: the fused form of PySurfaceResizer -> PySurfaceConverter."""
    sw, sh, dw, dh = geom
    host = make_nv12(sw, sh, seed=geom[1] + geom[2])
    src = upload(vali, gpu, host, sw, sh)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    pf = vali.PixelFormat[dst_fmt]
    dst = vali.Surface.Make(pf, dw, dh, gpu)
    assert vali.PySurfacePreprocessor(gpu).Run(src, dst, cc) == (True, vali.TaskExecInfo.SUCCESS)
    got = download(vali, gpu, dst, np.uint8)
    # the chain on the GPU
    cur = src
    if (sw, sh) != (dw, dh):
        cur = vali.Surface.Make(vali.NV12, dw, dh, gpu)
        assert vali.PySurfaceResizer(vali.NV12, gpu, interpolation=vali.Interpolation.LINEAR).Run(src, cur)[0]
    ref = vali.Surface.Make(pf, dw, dh, gpu)
    assert vali.PySurfaceConverter(gpu).Run(cur, ref, cc)[0]
    assert np.array_equal(got, download(vali, gpu, ref, np.uint8))
    # and the oracle
    n12 = host if (sw, sh) == (dw, dh) else oracle.resize_surface(
        np.ascontiguousarray(host).reshape(-1), "NV12", sw, sh, dw, dh).reshape(dh * 3 // 2, dw)
    assert np.array_equal(got, oracle.nv12_to_rgb(n12, dw, dh, oracle.csc(1), dst_fmt).reshape(-1))
    # a non-identity normalisation cannot apply to an 8-bit destination
    assert vali.PySurfacePreprocessor(gpu, mean=MEAN, std=STD).Run(src, dst, cc) == (False, vali.TaskExecInfo.NOT_SUPPORTED)
