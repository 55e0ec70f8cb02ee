"""Parity of the HIP NV12 -> RGB/BGR/RGB_PLANAR kernels against the CPU oracle: bit-exact.

Mirrors reference tests/test_PySurfaceConverter.py (:61-92 error code, :98-143 default
context sync+async, :228-300 NV12->RGB) with self-generated inputs (the reference's
test.nv12 / test.rgb blobs are missing).
"""
import numpy as np
import pytest

from conftest import GOLDEN, make_nv12

pytestmark = pytest.mark.gpu

VARIANTS = {
    "709_mpeg": ("BT_709", "MPEG", 1),
    "709_jpeg": ("BT_709", "JPEG", 2),
    "601_jpeg": ("BT_601", "JPEG", 0),
}


def convert(vali, gpu, nv12, w, h, dst_name, cc, is_async=False, cvt=None):
    src = vali.Surface.Make(vali.NV12, w, h, gpu)
    dst = vali.Surface.Make(vali.PixelFormat[dst_name], w, h, gpu)
    ok, info = vali.PyFrameUploader(gpu).Run(nv12.reshape(-1), src)
    assert ok, info
    cvt = cvt or vali.PySurfaceConverter(gpu)
    if is_async:
        ok, info = cvt.RunAsync(src, dst, cc)
        ev = vali.CudaStreamEvent(cvt.Stream, gpu)
        ev.Record()
        ev.Wait()
    else:
        ok, info = cvt.Run(src, dst, cc)
    assert ok and info == vali.TaskExecInfo.SUCCESS
    out = np.zeros(dst.HostSize, np.uint8)
    ok, info = vali.PySurfaceDownloader(gpu).Run(dst, out)
    assert ok
    return out


@pytest.mark.parametrize("size", [(64, 48), (848, 464), (1920, 1080), (1024, 2), (4096 + 16, 4)])
@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("dst", ["RGB", "BGR", "RGB_PLANAR"])
def test_bit_exact_fast_path(vali, gpu, oracle, size, variant, dst):
    w, h = size
    space, rng, ov = VARIANTS[variant]
    cc = vali.ColorspaceConversionContext(vali.ColorSpace[space], vali.ColorRange[rng])
    for seed in (-1, 0):
        nv12 = make_nv12(w, h, seed)
        got = convert(vali, gpu, nv12, w, h, dst, cc)
        want = oracle.nv12_to_rgb(nv12, w, h, oracle.csc(ov), dst).reshape(-1)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("size", [(50, 34), (424, 232), (18, 2), (2, 2), (1918, 1078), (34, 8)])
@pytest.mark.parametrize("dst", ["RGB", "BGR", "RGB_PLANAR"])
def test_bit_exact_ragged_sizes(vali, gpu, oracle, size, dst):
    """Widths that are not a multiple of 16 and odd heights take the byte-granular path."""
    w, h = size
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    nv12 = make_nv12(w, h, 3)
    got = convert(vali, gpu, nv12, w, h, dst, cc)
    want = oracle.nv12_to_rgb(nv12, w, h, oracle.csc(1), dst).reshape(-1)
    assert np.array_equal(got, want)


def test_real_video_frames(vali, gpu, oracle):
    """Two real 424x232 NV12 frames from the reference's tests/data/test_small.nv12."""
    raw = np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8)
    w, h = 424, 232
    frames = raw.reshape(2, h * 3 // 2, w)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    for f in frames:
        f = np.ascontiguousarray(f)
        got = convert(vali, gpu, f, w, h, "RGB", cc)
        assert np.array_equal(got, oracle.nv12_to_rgb(f, w, h, oracle.csc(1), "RGB").reshape(-1))


@pytest.mark.parametrize("is_async", [True, False])
def test_no_cc_ctx(vali, gpu, oracle, is_async):
    """Default context = BT.709 + JPEG (reference test_no_cc_ctx, TaskConvertSurface.cpp:117-118)."""
    w, h = 640, 360
    nv12 = make_nv12(w, h, 5)
    got = convert(vali, gpu, nv12, w, h, "RGB", None, is_async)
    assert np.array_equal(got, oracle.nv12_to_rgb(nv12, w, h, oracle.csc(2), "RGB").reshape(-1))


def test_unsupported_params(vali, gpu):
    """NV12 -> RGB with BT.601 + MPEG => UNSUPPORTED_FMT_CONV_PARAMS
    (reference tests/test_PySurfaceConverter.py:61-92)."""
    src = vali.Surface.Make(vali.NV12, 64, 48, gpu)
    dst = vali.Surface.Make(vali.RGB, 64, 48, gpu)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_601, vali.ColorRange.MPEG)
    ok, info = vali.PySurfaceConverter(gpu).Run(src, dst, cc)
    assert not ok and info == vali.TaskExecInfo.UNSUPPORTED_FMT_CONV_PARAMS


def test_size_mismatch_and_unsupported_pair(vali, gpu):
    src = vali.Surface.Make(vali.NV12, 64, 48, gpu)
    dst = vali.Surface.Make(vali.RGB, 32, 48, gpu)
    ok, info = vali.PySurfaceConverter(gpu).Run(src, dst)
    assert not ok and info == vali.TaskExecInfo.INVALID_INPUT      # TaskConvertSurface.cpp:1013-1015
    with pytest.raises(ValueError):                                 # :1085-1089
        vali.PySurfaceConverter(gpu).Run(src, vali.Surface.Make(vali.YUV422, 64, 48, gpu))


@pytest.mark.parametrize("dst", ["RGB", "RGB_PLANAR"])
def test_batch_equals_single(vali, gpu, oracle, dst):
    w, h, n = 1920, 1080, 6
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    cvt = vali.PySurfaceConverter(gpu)
    upl, dwn = vali.PyFrameUploader(gpu), vali.PySurfaceDownloader(gpu)
    frames = [make_nv12(w, h, s) for s in range(n)]
    srcs = [vali.Surface.Make(vali.NV12, w, h, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(vali.PixelFormat[dst], w, h, gpu) for _ in range(n)]
    for f, s in zip(frames, srcs):
        assert upl.Run(f.reshape(-1), s)[0]
    ok, info = cvt.RunBatch(srcs, dsts, cc)
    assert ok and info == vali.TaskExecInfo.SUCCESS
    for f, d in zip(frames, dsts):
        out = np.zeros(d.HostSize, np.uint8)
        assert dwn.Run(d, out)[0]
        assert np.array_equal(out, oracle.nv12_to_rgb(f, w, h, oracle.csc(1), dst).reshape(-1))


def test_full_size_properties_2160p(vali, gpu, oracle):
    """BASELINE full size: a checksum over a 2160p batch equals the checksum of the oracle on
    the distinct inputs, and conversion is idempotent across repeated launches."""
    w, h, n = 3840, 2160, 4
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    cvt = vali.PySurfaceConverter(gpu)
    upl, dwn = vali.PyFrameUploader(gpu), vali.PySurfaceDownloader(gpu)
    frames = [make_nv12(w, h, s, full_range=False) for s in (-1, 1)]
    srcs = [vali.Surface.Make(vali.NV12, w, h, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(vali.RGB, w, h, gpu) for _ in range(n)]
    for i, s in enumerate(srcs):
        assert upl.Run(frames[i % 2].reshape(-1), s)[0]
    batch = cvt.PrepareBatch(srcs, dsts)
    outs = []
    for _ in range(2):
        assert cvt.RunBatch(batch, cc_ctx=cc)[0]
        cur = []
        for d in dsts:
            o = np.zeros(d.HostSize, np.uint8)
            assert dwn.Run(d, o)[0]
            cur.append(o)
        outs.append(cur)
    wants = [oracle.nv12_to_rgb(f, w, h, oracle.csc(1), "RGB").reshape(-1) for f in frames]
    for i in range(n):
        assert np.array_equal(outs[0][i], wants[i % 2])
        assert np.array_equal(outs[0][i], outs[1][i])


def test_foreign_pitch_via_dlpack(vali, gpu, oracle):
    """A torch tensor with an odd row stride (unaligned rows) goes through from_dlpack and
    the generic kernel path; result still bit-exact."""
    import torch

    w, h = 64, 32
    nv12 = make_nv12(w, h, 9)
    buf = torch.zeros((h * 3 // 2, w + 3), dtype=torch.uint8, device="cuda")
    view = buf[:, :w]
    view.copy_(torch.from_numpy(nv12))
    src = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(view), vali.NV12)
    assert src.Pitch == w + 3 and not src.IsOwnMemory and src.Width == w and src.Height == h
    dst = vali.Surface.Make(vali.RGB, w, h, gpu)
    torch.cuda.synchronize()
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    assert vali.PySurfaceConverter(gpu).Run(src, dst, cc)[0]
    out = np.zeros(dst.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(dst, out)[0]
    assert np.array_equal(out, oracle.nv12_to_rgb(nv12, w, h, oracle.csc(1), "RGB").reshape(-1))
