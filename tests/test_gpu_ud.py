"""PySurfaceUD on the GPU vs the CPU oracle: bit-exact for u8/u16 AND for the float outputs
(same operations in the same order).  Mirrors reference tests/test_PySurfaceUD.py:135-188
(NV12 / P10 sources -> every supported dst format at 640x360) with synthetic frames."""
import numpy as np
import pytest

from conftest import make_nv12

pytestmark = pytest.mark.gpu

NV12_DSTS = ["YUV444", "RGB", "RGB_PLANAR", "RGB_32F", "RGB_32F_PLANAR"]
P10_DSTS = ["YUV444_10bit", "RGB_32F", "RGB_32F_PLANAR"]
DTYPE = {"YUV444": np.uint8, "YUV444_10bit": np.uint16, "RGB": np.uint8, "RGB_PLANAR": np.uint8,
         "RGB_32F": np.float32, "RGB_32F_PLANAR": np.float32}


def make_p10(w, h, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 1024, (h * 3 // 2, w), dtype=np.uint16) << 6).astype(np.uint16)


def run_ud(vali, gpu, host, sw, sh, src_fmt, dw, dh, dst_fmt, is_async=False):
    src = vali.Surface.Make(vali.PixelFormat[src_fmt], sw, sh, gpu)
    dst = vali.Surface.Make(vali.PixelFormat[dst_fmt], dw, dh, gpu)
    assert vali.PyFrameUploader(gpu).Run(host.reshape(-1).view(np.uint8), src)[0]
    ud = vali.PySurfaceUD(gpu)
    ok, info = ud.RunAsync(src, dst) if is_async else ud.Run(src, dst)
    assert ok and info == vali.TaskExecInfo.SUCCESS
    if is_async:
        ev = vali.CudaStreamEvent(ud.Stream, gpu)
        ev.Record()
        ev.Wait()
    out = np.zeros(dst.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(dst, out)[0]
    return out.view(DTYPE[dst_fmt])


@pytest.mark.parametrize("dst", NV12_DSTS)
@pytest.mark.parametrize("geom", [(848, 464, 640, 360), (1920, 1080, 960, 540), (640, 360, 1280, 720),
                                  (424, 232, 421, 233), (64, 48, 7, 5)])
def test_nv12_sources(vali, gpu, oracle, dst, geom):
    sw, sh, dw, dh = geom
    nv = make_nv12(sw, sh, 11)
    got = run_ud(vali, gpu, nv, sw, sh, "NV12", dw, dh, dst)
    want = oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, dst).reshape(-1)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("dst", P10_DSTS)
@pytest.mark.parametrize("is_async", [False, True])
def test_p10_sources(vali, gpu, oracle, dst, is_async):
    sw, sh, dw, dh = 848, 464, 640, 360
    p10 = make_p10(sw, sh, 5)
    got = run_ud(vali, gpu, p10, sw, sh, "P10", dw, dh, dst, is_async)
    want = oracle.ud_nv12(p10, sw, sh, "P10", dw, dh, dst).reshape(-1)
    assert np.array_equal(got, want)


def test_not_supported_pairs(vali, gpu):
    """reference UDSurface.cpp:137-149: unlisted pair -> NOT_SUPPORTED (no exception)."""
    ud = vali.PySurfaceUD(gpu)
    src = vali.Surface.Make(vali.RGB, 64, 48, gpu)
    dst = vali.Surface.Make(vali.RGB, 32, 24, gpu)
    assert ud.Run(src, dst) == (False, vali.TaskExecInfo.NOT_SUPPORTED)
    assert (vali.NV12, vali.RGB_32F_PLANAR) in vali.PySurfaceUD.SupportedFormats()
    assert len(vali.PySurfaceUD.SupportedFormats()) == 10


def test_batch_2160p_to_1080p(vali, gpu, oracle):
    """BASELINE config 4, first half: NV12 2160p -> RGB 1080p; full size, checked on crops
    of the oracle (the oracle on a full 2160p frame takes seconds, so 2 frames only)."""
    sw, sh, dw, dh, n = 3840, 2160, 1920, 1080, 2
    ud = vali.PySurfaceUD(gpu)
    frames = [make_nv12(sw, sh, s, full_range=False) for s in range(n)]
    srcs = [vali.Surface.Make(vali.NV12, sw, sh, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(vali.RGB, dw, dh, gpu) for _ in range(n)]
    for f, s in zip(frames, srcs):
        assert vali.PyFrameUploader(gpu).Run(f.reshape(-1), s)[0]
    assert ud.RunBatch(srcs, dsts) == (True, vali.TaskExecInfo.SUCCESS)
    for f, d in zip(frames, dsts):
        out = np.zeros(d.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(d, out)[0]
        assert np.array_equal(out, oracle.ud_nv12(f, sw, sh, "NV12", dw, dh, "RGB").reshape(-1))


@pytest.mark.parametrize("src_fmt,dst_fmt,dt", [("YUV420", "YUV444", np.uint8),
                                                ("YUV420_10bit", "YUV444_10bit", np.uint16)])
def test_planar_sources(vali, gpu, oracle, src_fmt, dst_fmt, dt):
    """reference tests/test_PySurfaceUD.py:71-132 (CPU-decoded planar sources)."""
    sw, sh, dw, dh = 848, 464, 640, 360
    rng = np.random.default_rng(8)
    src = vali.Surface.Make(vali.PixelFormat[src_fmt], sw, sh, gpu)
    host = (rng.random(src.HostSize // np.dtype(dt).itemsize) * 1000).astype(dt)
    assert vali.PyFrameUploader(gpu).Run(host.view(np.uint8), src)[0]
    dst = vali.Surface.Make(vali.PixelFormat[dst_fmt], dw, dh, gpu)
    assert vali.PySurfaceUD(gpu).Run(src, dst) == (True, vali.TaskExecInfo.SUCCESS)
    out = np.zeros(dst.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(dst, out)[0]
    got = out.view(dt)
    off, want = 0, []
    for pw, ph in ((sw, sh), (sw // 2, sh // 2), (sw // 2, sh // 2)):
        plane = np.ascontiguousarray(host[off: off + pw * ph].reshape(ph, pw))
        want.append(oracle.resize_plane(plane, 1, dw, dh).reshape(-1))
        off += pw * ph
    assert np.array_equal(got, np.concatenate(want))


# ---- fused UD + quarter-turn rotation (BASELINE config 4 as one pass) ---------------------------
@pytest.mark.parametrize("angle", [90.0, 180.0, 270.0, -90.0])
@pytest.mark.parametrize("geom", [(3840, 2160, 1920, 1080), (848, 464, 640, 360), (640, 360, 1280, 720),
                                  (424, 232, 421, 233), (64, 48, 7, 5), (130, 70, 58, 34), (1920, 1080, 250, 251)])
def test_ud_rotated_equals_ud_then_rotator(vali, gpu, oracle, angle, geom):
    sw, sh, uw, uh = geom                       # uw x uh = size of the un-rotated UD output
    nv = make_nv12(sw, sh, 23)
    src = vali.Surface.Make(vali.NV12, sw, sh, gpu)
    assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), src)[0]
    odd = int(round(angle / 90.0)) % 2 == 1
    dw, dh = (uh, uw) if odd else (uw, uh)
    ud = vali.PySurfaceUD(gpu)
    fused = vali.Surface.Make(vali.RGB, dw, dh, gpu)
    assert ud.RunRotated(src, fused, angle) == (True, vali.TaskExecInfo.SUCCESS)
    # the chain it replaces, with this library's own tasks
    mid = vali.Surface.Make(vali.RGB, uw, uh, gpu)
    chain = vali.Surface.Make(vali.RGB, dw, dh, gpu)
    assert ud.Run(src, mid)[0] and vali.PySurfaceRotator(gpu).Run(mid, chain, angle)[0]
    a = np.zeros(fused.HostSize, np.uint8)
    b = np.zeros(chain.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(fused, a)[0] and vali.PySurfaceDownloader(gpu).Run(chain, b)[0]
    assert np.array_equal(a, b)
    # and the CPU oracle: UD, then the exact permutation of a quarter turn
    want = oracle.ud_nv12(nv, sw, sh, "NV12", uw, uh, "RGB").reshape(uh, uw, 3)
    k = int(round(angle / 90.0)) % 4            # rotate.hip: 90 deg = dst(x', y') = src(W-1-y', x')
    want = np.rot90(want, k=k)                  # np.rot90 (counter-clockwise) is that permutation
    assert np.array_equal(a.reshape(dh, dw, 3), want)


def test_ud_rotated_batch_and_errors(vali, gpu, oracle):
    sw, sh, uw, uh, n = 1280, 720, 640, 360, 5
    nv = make_nv12(sw, sh, 29)
    srcs = [vali.Surface.Make(vali.NV12, sw, sh, gpu) for _ in range(n)]
    for s_ in srcs:
        assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), s_)[0]
    dsts = [vali.Surface.Make(vali.RGB, uh, uw, gpu) for _ in range(n)]
    ud = vali.PySurfaceUD(gpu)
    assert ud.RunRotatedBatch(srcs, dsts, angle=270.0) == (True, vali.TaskExecInfo.SUCCESS)
    want = np.rot90(oracle.ud_nv12(nv, sw, sh, "NV12", uw, uh, "RGB").reshape(uh, uw, 3), k=3)
    for d in dsts:
        out = np.zeros(d.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(d, out)[0]
        assert np.array_equal(out.reshape(uw, uh, 3), want)
    assert ud.RunRotated(srcs[0], dsts[0], 45.0) == (False, vali.TaskExecInfo.NOT_SUPPORTED)
    assert ud.RunRotated(srcs[0], dsts[0], 0.0) == (False, vali.TaskExecInfo.NOT_SUPPORTED)
    planar = vali.Surface.Make(vali.RGB_PLANAR, uh, uw, gpu)
    assert ud.RunRotated(srcs[0], planar, 90.0) == (False, vali.TaskExecInfo.NOT_SUPPORTED)


@pytest.mark.parametrize("dst", ["YUV444_10bit", "RGB_32F", "RGB_32F_PLANAR"])
@pytest.mark.parametrize("geom", [(1920, 1080, 960, 540), (1280, 720, 300, 170), (640, 360, 1280, 720),
                                  (2048, 64, 1000, 30), (130, 70, 58, 34)])
def test_p10_geometries(vali, gpu, oracle, dst, geom):
    """16-bit sources: 2x downscale (two staging chunks per lane), > 4x (gather path), upscale, ragged."""
    sw, sh, dw, dh = geom
    rng = np.random.default_rng(sw + dh)
    p10 = (rng.integers(0, 1024, (sh * 3 // 2, sw), dtype=np.uint16) << 6).astype(np.uint16)
    got = run_ud(vali, gpu, p10, sw, sh, "P10", dw, dh, dst)
    want = oracle.ud_nv12(p10, sw, sh, "P10", dw, dh, dst).reshape(-1)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
