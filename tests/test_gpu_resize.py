"""PySurfaceResizer on the GPU vs the oracle (bit-exact; float32 too).

Mirrors reference tests/test_PySurfaceResizer.py:64-140 (NV12 848x464 -> 424x232) with the two
real NV12 frames that exist offline, plus the other formats ResizeSurface accepts
(src/TC/src/TaskResizeSurface.cpp:293-309)."""
import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

DT = {"RGB_32F": np.float32, "RGB_32F_PLANAR": np.float32, "P10": np.uint16,
      "YUV444_10bit": np.uint16, "YUV420_10bit": np.uint16}


def roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, is_async=False, interp=None):
    pf = vali.PixelFormat[fmt]
    src = vali.Surface.Make(pf, sw, sh, gpu)
    dst = vali.Surface.Make(pf, dw, dh, gpu)
    assert vali.PyFrameUploader(gpu).Run(host.view(np.uint8), src)[0]
    # (the task's own default is the reference's filter, Lanczos: test_default_filter_is_the_references)
    rs = vali.PySurfaceResizer(pf, gpu, interpolation=vali.Interpolation.LINEAR if interp is None else interp)
    ok, info = rs.RunAsync(src, dst) if is_async else rs.Run(src, dst)
    assert ok and info == vali.TaskExecInfo.SUCCESS
    if is_async:
        ev = vali.CudaStreamEvent(rs.Stream, gpu)
        ev.Record()
        ev.Wait()
    out = np.zeros(dst.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(dst, out)[0]
    return out.view(host.dtype)


@pytest.mark.parametrize("fmt", ["NV12", "YUV420", "YUV444", "RGB", "BGR", "RGB_PLANAR", "RGB_32F",
                                 "RGB_32F_PLANAR", "Y", "P10", "YUV422", "YUV444_10bit"])
@pytest.mark.parametrize("geom", [(848, 464, 424, 232), (640, 360, 1000, 500), (130, 70, 58, 34),
                                  (1920, 1080, 1280, 720)])
def test_resize_bit_exact(vali, gpu, oracle, fmt, geom):
    sw, sh, dw, dh = geom
    dt = DT.get(fmt, np.uint8)
    n = vali.Surface.Make(vali.PixelFormat[fmt], sw, sh, gpu).HostSize // np.dtype(dt).itemsize
    rng = np.random.default_rng(21)
    host = (rng.random(n) * (1000 if dt == np.uint16 else 255)).astype(dt)
    got = roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh)
    want = oracle.resize_surface(host, fmt, sw, sh, dw, dh)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("is_async", [True, False])
def test_resize_nv12_real_frames(vali, gpu, oracle, is_async):
    raw = np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8).reshape(2, -1)
    for frame in raw:
        got = roundtrip(vali, gpu, "NV12", frame.copy(), 424, 232, 212, 116, is_async)
        assert np.array_equal(got, oracle.resize_surface(frame, "NV12", 424, 232, 212, 116))


def test_default_filter_is_the_references(vali, gpu, oracle):
    """PySurfaceResizer(format, gpu_id) -- the reference's constructor -- resizes with the reference's only
    filter, Lanczos (NPPI_INTER_LANCZOS at every call site, TaskResizeSurface.cpp:67,116,224,273)."""
    assert vali.PySurfaceResizer(vali.NV12, gpu).Interpolation == vali.Interpolation.LANCZOS
    sw, sh, dw, dh = 848, 464, 640, 360
    rng = np.random.default_rng(12)
    host = rng.integers(0, 256, sw * sh * 3 // 2, dtype=np.uint8)
    src = vali.Surface.Make(vali.NV12, sw, sh, gpu)
    dst = vali.Surface.Make(vali.NV12, dw, dh, gpu)
    assert vali.PyFrameUploader(gpu).Run(host, src)[0]
    assert vali.PySurfaceResizer(vali.NV12, gpu).Run(src, dst) == (True, vali.TaskExecInfo.SUCCESS)
    out = np.zeros(dst.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(dst, out)[0]
    assert np.array_equal(out, oracle.resize_surface(host, "NV12", sw, sh, dw, dh, "lanczos"))
    assert not np.array_equal(out, oracle.resize_surface(host, "NV12", sw, sh, dw, dh, "linear"))


def test_resize_errors(vali, gpu):
    with pytest.raises(RuntimeError):                      # TaskResizeSurface.cpp:307-308
        vali.PySurfaceResizer(vali.PixelFormat.GRAY12, gpu)
    rs = vali.PySurfaceResizer(vali.NV12, gpu)
    a = vali.Surface.Make(vali.NV12, 64, 48, gpu)
    b = vali.Surface.Make(vali.YUV420, 32, 24, gpu)
    assert rs.Run(a, b) == (False, vali.TaskExecInfo.INVALID_INPUT)   # :46-48


def test_resize_batch_2160p_to_720p(vali, gpu, oracle):
    """BASELINE config 3 geometry (batch reduced to 4 for the oracle's sake)."""
    sw, sh, dw, dh, n = 3840, 2160, 1280, 720, 4
    rs = vali.PySurfaceResizer(vali.NV12, gpu, interpolation=vali.Interpolation.LINEAR)
    rng = np.random.default_rng(5)
    frames = [rng.integers(0, 256, sw * sh * 3 // 2, dtype=np.uint8) for _ in range(2)]
    srcs = [vali.Surface.Make(vali.NV12, sw, sh, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(vali.NV12, dw, dh, gpu) for _ in range(n)]
    for i, s in enumerate(srcs):
        assert vali.PyFrameUploader(gpu).Run(frames[i % 2], s)[0]
    assert rs.RunBatch(srcs, dsts) == (True, vali.TaskExecInfo.SUCCESS)
    wants = [oracle.resize_surface(f, "NV12", sw, sh, dw, dh) for f in frames]
    for i, d in enumerate(dsts):
        out = np.zeros(d.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(d, out)[0]
        assert np.array_equal(out, wants[i % 2])


# ---- Lanczos-3 (the reference's NPPI_INTER_LANCZOS, restated) --------------------------------
@pytest.mark.parametrize("fmt", ["NV12", "YUV420", "RGB", "RGB_PLANAR", "RGB_32F", "Y", "P10", "YUV444_10bit"])
@pytest.mark.parametrize("geom", [(848, 464, 424, 232), (640, 360, 1000, 500), (130, 70, 58, 34),
                                  (1920, 1080, 1280, 720), (64, 48, 640, 480)])
def test_lanczos_bit_exact(vali, gpu, oracle, fmt, geom):
    sw, sh, dw, dh = geom
    dt = DT.get(fmt, np.uint8)
    n = vali.Surface.Make(vali.PixelFormat[fmt], sw, sh, gpu).HostSize // np.dtype(dt).itemsize
    rng = np.random.default_rng(22)
    host = (rng.random(n) * (1000 if dt == np.uint16 else 255)).astype(dt)
    got = roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=vali.Interpolation.LANCZOS)
    want = oracle.resize_surface(host, fmt, sw, sh, dw, dh, "lanczos")
    assert np.array_equal(got, want)


# ---- bicubic (NPPI_INTER_CUBIC; the filter BASELINE.json's north_star names next to bilinear) ---
@pytest.mark.parametrize("fmt", ["NV12", "YUV420", "RGB", "RGB_PLANAR", "RGB_32F", "Y", "P10", "YUV444_10bit"])
@pytest.mark.parametrize("geom", [(848, 464, 424, 232), (640, 360, 1000, 500), (130, 70, 58, 34),
                                  (1920, 1080, 1280, 720), (64, 48, 640, 480)])
def test_cubic_bit_exact(vali, gpu, oracle, fmt, geom):
    sw, sh, dw, dh = geom
    dt = DT.get(fmt, np.uint8)
    n = vali.Surface.Make(vali.PixelFormat[fmt], sw, sh, gpu).HostSize // np.dtype(dt).itemsize
    rng = np.random.default_rng(23)
    host = (rng.random(n) * (1000 if dt == np.uint16 else 255)).astype(dt)
    got = roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=vali.Interpolation.CUBIC)
    want = oracle.resize_surface(host, fmt, sw, sh, dw, dh, "cubic")
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def test_cubic_2160p_noninteger_and_wide_span(vali, gpu, oracle):
    """1080x608 from 2160p: the luma span fits the LDS strip, a 3.55x chroma span does not (gather)."""
    sw, sh = 3840, 2160
    rng = np.random.default_rng(8)
    frame = rng.integers(0, 256, sw * sh * 3 // 2, dtype=np.uint8)
    for dw, dh in ((1080, 608), (2000, 1126)):
        got = roundtrip(vali, gpu, "NV12", frame, sw, sh, dw, dh, interp=vali.Interpolation.CUBIC)
        assert np.array_equal(got, oracle.resize_surface(frame, "NV12", sw, sh, dw, dh, "cubic"))


def test_lanczos_real_frames_match_reference_fixture_geometry(vali, gpu, oracle):
    """reference tests/test_PySurfaceResizer.py resizes by exactly 2x with NPP Lanczos; at integer
    factors an interpolating kernel is the point sample src[2y][2x] -- same as bilinear."""
    raw = np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8).reshape(2, -1)
    for frame in raw:
        lz = roundtrip(vali, gpu, "NV12", frame.copy(), 424, 232, 212, 116, interp=vali.Interpolation.LANCZOS)
        bl = roundtrip(vali, gpu, "NV12", frame.copy(), 424, 232, 212, 116)
        assert np.array_equal(lz, bl)
        assert np.array_equal(lz, oracle.resize_surface(frame, "NV12", 424, 232, 212, 116, "lanczos"))


def test_lanczos_batch_2160p_to_720p(vali, gpu, oracle):
    sw, sh, dw, dh, n = 3840, 2160, 1280, 720, 3
    rs = vali.PySurfaceResizer(vali.NV12, gpu, interpolation=vali.Interpolation.LANCZOS)
    assert rs.Interpolation == vali.Interpolation.LANCZOS
    rng = np.random.default_rng(6)
    frame = rng.integers(0, 256, sw * sh * 3 // 2, dtype=np.uint8)
    srcs = [vali.Surface.Make(vali.NV12, sw, sh, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(vali.NV12, dw, dh, gpu) for _ in range(n)]
    for s_ in srcs:
        assert vali.PyFrameUploader(gpu).Run(frame, s_)[0]
    assert rs.RunBatch(srcs, dsts) == (True, vali.TaskExecInfo.SUCCESS)
    want = oracle.resize_surface(frame, "NV12", sw, sh, dw, dh, "lanczos")
    for d in dsts:
        out = np.zeros(d.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(d, out)[0]
        assert np.array_equal(out, want)


def test_lanczos_noninteger_2160p_to_1080x608(vali, gpu, oracle):
    """wide spans: the chroma plane of a 3.55x downscale exceeds the LDS strip (gather path)."""
    sw, sh, dw, dh = 3840, 2160, 1080, 608
    rng = np.random.default_rng(7)
    frame = rng.integers(0, 256, sw * sh * 3 // 2, dtype=np.uint8)
    got = roundtrip(vali, gpu, "NV12", frame, sw, sh, dw, dh, interp=vali.Interpolation.LANCZOS)
    assert np.array_equal(got, oracle.resize_surface(frame, "NV12", sw, sh, dw, dh, "lanczos"))


@pytest.mark.parametrize("interp", ["linear", "cubic", "lanczos"])
def test_float_planes_keep_denormals_and_signed_zero(vali, gpu, oracle, interp):
    """f32 planes: subnormal inputs, negative values and -0.0 go through the same IEEE arithmetic as
    on the CPU (no flush-to-zero in the kernels): bit-exact, not merely close."""
    sw, sh, dw, dh = 96, 40, 151, 67
    rng = np.random.default_rng(9)
    host = rng.random(sw * sh * 3, dtype=np.float32)
    host[::7] *= np.float32(1e-41)                 # subnormals
    host[::11] *= np.float32(-1.0)
    host[::13] = np.float32(-0.0)
    host[::17] = np.float32(3.0e38)                # near FLT_MAX: products overflow to inf identically
    it = vali.Interpolation[interp.upper()]
    got = roundtrip(vali, gpu, "RGB_32F", host, sw, sh, dw, dh, interp=it)
    want = oracle.resize_surface(host, "RGB_32F", sw, sh, dw, dh, interp)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---- integer scale factors: the point-sample form of the kernel (BASELINE config 3 is an exact 3x) ----
@pytest.mark.parametrize("fmt", ["NV12", "YUV420", "YUV444", "RGB", "RGB_PLANAR", "Y", "P10", "YUV422",
                                 "YUV444_10bit", "RGB_32F"])
@pytest.mark.parametrize("geom", [(3840, 2160, 1280, 720), (1920, 1080, 960, 540), (1200, 64, 100, 16),   # 3x, 2x, 12x4
                                  (640, 360, 640, 360), (1002, 36, 334, 18), (4096, 32, 256, 8),           # 1x, ragged 3x2, 16x4
                                  (1280, 720, 1280, 360), (84, 40, 42, 40)])                               # one axis only
def test_integer_scale_is_the_point_sample(vali, gpu, oracle, fmt, geom):
    sw, sh, dw, dh = geom
    dt = DT.get(fmt, np.uint8)
    n = vali.Surface.Make(vali.PixelFormat[fmt], sw, sh, gpu).HostSize // np.dtype(dt).itemsize
    rng = np.random.default_rng(31)
    host = (rng.random(n) * (1023 if dt == np.uint16 else 255)).astype(dt)
    want = oracle.resize_surface(host, fmt, sw, sh, dw, dh, "linear")
    for interp in (vali.Interpolation.LINEAR, vali.Interpolation.CUBIC, vali.Interpolation.LANCZOS):
        if dt == np.float32 and interp != vali.Interpolation.LINEAR:
            continue  # float planes keep each filter's own arithmetic
        got = roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=interp)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), interp
    # and it IS the decimated source, plane by plane
    if fmt == "Y":
        src = host.reshape(sh, sw)
        assert np.array_equal(want.reshape(dh, dw), src[:: sh // dh, :: sw // dw])


def test_integer_scale_foreign_unaligned_source(vali, gpu, oracle):
    """unaligned rows leave the LDS-staged path: the point-sample kernel falls back to its gather"""
    from test_gpu_edge_geometry import download, foreign_nv12
    from conftest import make_nv12
    sw, sh, dw, dh = 636, 180, 212, 60
    nv = make_nv12(sw, sh, 11)
    for pad, skew in ((3, 0), (16, 5)):
        src, keep = foreign_nv12(vali, sw, sh, nv, pad, skew)
        small = vali.Surface.Make(vali.NV12, dw, dh, gpu)
        assert vali.PySurfaceResizer(vali.NV12, gpu).Run(src, small)[0]
        assert np.array_equal(download(vali, gpu, small), oracle.resize_surface(nv.reshape(-1), "NV12", sw, sh, dw, dh, "linear"))
        del keep


def test_lanczos_batch_large_enough_for_32_row_waves(vali, gpu, oracle):
    """A launch with >= 1024 tiles of 128 rows takes the 32-rows-per-wave form of the Lanczos kernel by itself (the
    form the batched benchmarks run); every frame bit-exact."""
    sw, sh, dw, dh, n = 640, 360, 854, 480, 48
    rs = vali.PySurfaceResizer(vali.NV12, gpu)
    rng = np.random.default_rng(77)
    frames = [rng.integers(0, 256, sw * sh * 3 // 2, dtype=np.uint8) for _ in range(3)]
    srcs = [vali.Surface.Make(vali.NV12, sw, sh, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(vali.NV12, dw, dh, gpu) for _ in range(n)]
    for i, s in enumerate(srcs):
        assert vali.PyFrameUploader(gpu).Run(frames[i % 3], s)[0]
    assert rs.RunBatch(srcs, dsts) == (True, vali.TaskExecInfo.SUCCESS)
    wants = [oracle.resize_surface(f, "NV12", sw, sh, dw, dh, "lanczos") for f in frames]
    for i, d in enumerate(dsts):
        out = np.zeros(d.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(d, out)[0]
        assert np.array_equal(out, wants[i % 3]), i


# ---- the columns-first kernels (planes that shrink vertically, vali_amd/csrc/resize_cols.hip) -------------------
# geometry -> what it reaches: slot counts 3 / 4 / 6 (Lanczos) and 2 / 3 / 4 (bicubic) by the vertical ratio, the 2:1-along-x
# form (src_w == 2 dst_w) and the general one, ragged last tiles, planes narrower than one lane's 8 elements (direct form),
# a vertical shrink with a horizontal stretch, rows kept (src_h == dst_h), a plane pair that takes BOTH orders (YUV420 ->
# taller: luma shrinks, chroma ... ) is covered by the planar UD tests.
COLS_GEOMS = [(1920, 1080, 1280, 720),    # 3:2 along x AND y: the uniform-weight form, 4 slots
              (1536, 600, 1024, 280),     # 3:2 along x, 2.14 down the rows: 3 slots
              (1488, 404, 992, 400),      # 3:2 along x (two waves per row + a partial one: 992 = 2 x 496), 6 slots
              (24, 100, 16, 50),          # 3:2, one lane of outputs per plane row (chroma: 12 -> 8 elements)
              (1280, 722, 640, 364),      # x2, ratio 1.98: 4 slots (3 bicubic)
              (1280, 720, 640, 238),      # x2, ratio 3.03: 3 slots (2)
              (636, 364, 318, 310),       # x2, ratio 1.17: 6 slots (4), ragged tile
              (1282, 360, 640, 250),      # general, 2.003 along x
              (1000, 700, 1400, 300),     # shrink rows, stretch columns
              (700, 540, 334, 540),       # rows kept: every row weight is 0 or 1
              (12, 300, 6, 100),          # narrow planes: the direct form for the chroma planes / x2 refused
              (2600, 800, 90, 64)]        # 29:1 along x, 12.5:1 along y: tiles of a few elements, sparse rows


@pytest.mark.parametrize("fmt", ["NV12", "YUV420", "RGB", "RGB_32F", "Y", "P10", "YUV444_10bit"])
@pytest.mark.parametrize("geom", COLS_GEOMS)
@pytest.mark.parametrize("interp", ["lanczos", "cubic"])
def test_columns_first_forms_bit_exact(vali, gpu, oracle, fmt, geom, interp):
    sw, sh, dw, dh = geom
    dt = DT.get(fmt, np.uint8)
    n = vali.Surface.Make(vali.PixelFormat[fmt], sw, sh, gpu).HostSize // np.dtype(dt).itemsize
    rng = np.random.default_rng(sw * 7 + dh)
    host = (rng.random(n) * (1000 if dt == np.uint16 else 255)).astype(dt)
    mode = vali.Interpolation.LANCZOS if interp == "lanczos" else vali.Interpolation.CUBIC
    want = oracle.resize_surface(host, fmt, sw, sh, dw, dh, interp)
    assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=mode), want)
    with vali.tuning.Override(RESIZE_POINT=0):           # the general columns-first form where the 2:1 one applies
        assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=mode), want)
    with vali.tuning.Override(RESIZE_FORCE_GATHER=1):    # one output element per thread
        assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=mode), want)


@pytest.mark.parametrize("rows_mode", [0, 1, 2, 3])
def test_columns_first_batch_and_rows_per_wave(vali, gpu, oracle, rows_mode):
    """a batch (every frame through the same launch) under every rows-per-wave form; float planes skip idle slots
    instead of multiplying by zero: an infinity next to the window must not leak into it"""
    sw, sh, dw, dh, n = 1280, 722, 640, 364, 5
    rng = np.random.default_rng(77)
    frames = [rng.integers(0, 256, sw * sh * 3 // 2, dtype=np.uint8) for _ in range(2)]
    srcs = [vali.Surface.Make(vali.NV12, sw, sh, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(vali.NV12, dw, dh, gpu) for _ in range(n)]
    for i, s in enumerate(srcs):
        assert vali.PyFrameUploader(gpu).Run(frames[i % 2], s)[0]
    with vali.tuning.Override(RESIZE_NO_SEPARABLE=rows_mode):
        assert vali.PySurfaceResizer(vali.NV12, gpu).RunBatch(srcs, dsts) == (True, vali.TaskExecInfo.SUCCESS)
    wants = [oracle.resize_surface(f, "NV12", sw, sh, dw, dh, "lanczos") for f in frames]
    for i, d in enumerate(dsts):
        out = np.zeros(d.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(d, out)[0]
        assert np.array_equal(out, wants[i % 2])
    # float plane with infinities and NaNs sprinkled in: bit-identical to the oracle, NaN payloads aside
    fw, fh = 400, 300
    host = rng.random(fw * fh * 3).astype(np.float32)
    host[rng.integers(0, host.size, 40)] = np.inf
    host[rng.integers(0, host.size, 10)] = -np.inf
    with vali.tuning.Override(RESIZE_NO_SEPARABLE=rows_mode):
        got = roundtrip(vali, gpu, "RGB_32F", host, fw, fh, 180, 140, interp=vali.Interpolation.LANCZOS)
    want = oracle.resize_surface(host, "RGB_32F", fw, fh, 180, 140, "lanczos")
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan) and np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32))


# ---- one-channel planes exactly doubled both ways (vali_amd/csrc/resize_up2.hip) -------------------------------------
# (source size): widths that are / are not multiples of 4 per plane (the second kind keeps the general kernel), one wave per
# row and several, 1 .. 3 rows (every tap clamped), heights around the rows-per-wave forms
UP2_SIZES = [(8, 2), (16, 6), (248, 20), (256, 64), (500, 37), (1000, 130), (960, 540), (36, 4), (1928, 70), (6, 10), (244, 12), (492, 8)]


@pytest.mark.parametrize("fmt", ["Y", "YUV420", "YUV444", "RGB_PLANAR", "YUV444_10bit", "NV12", "P10"])
@pytest.mark.parametrize("size", UP2_SIZES)
@pytest.mark.parametrize("interp", ["lanczos", "cubic"])
def test_doubled_planes_bit_exact(vali, gpu, oracle, fmt, size, interp):
    sw, sh = size
    if fmt in ("YUV420", "NV12", "P10") and (sw | sh) & 1:
        pytest.skip("4:2:0 needs even sizes")
    dt = DT.get(fmt, np.uint8)
    n = vali.Surface.Make(vali.PixelFormat[fmt], sw, sh, gpu).HostSize // np.dtype(dt).itemsize
    rng = np.random.default_rng(sw * 13 + sh)
    host = (rng.random(n) * (1023 if dt == np.uint16 else 255)).astype(dt)
    host[:: max(1, n // 50)] = np.iinfo(dt).max if dt != np.uint16 else 1023   # saturation / overshoot next to dark pixels
    if fmt == "P10":
        host = (host.astype(np.uint16) << 6).astype(np.uint16)
    mode = vali.Interpolation.LANCZOS if interp == "lanczos" else vali.Interpolation.CUBIC
    want = oracle.resize_surface(host, fmt, sw, sh, 2 * sw, 2 * sh, interp)
    assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, 2 * sw, 2 * sh, interp=mode), want)
    for rows_mode in (1, 2, 3):                            # 8 / 2 / 64 source rows per wave
        with vali.tuning.Override(RESIZE_NO_SEPARABLE=rows_mode):
            assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, 2 * sw, 2 * sh, interp=mode), want)
    with vali.tuning.Override(RESIZE_POINT=0):            # the general rows-first kernel
        assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, 2 * sw, 2 * sh, interp=mode), want)


@pytest.mark.parametrize("pair", [("YUV420", "YUV444", np.uint8), ("YUV420_10bit", "YUV444_10bit", np.uint16)])
@pytest.mark.parametrize("size", [(640, 360), (1920, 1080), (72, 34), (70, 34)])   # (70: chroma 35 wide, not a multiple of 4: the general kernel)
def test_planar_ud_at_unchanged_size_takes_the_plane_forms(vali, gpu, oracle, pair, size):
    """UDPlanar's everyday case: luma 1:1 (the point form = a copy), chroma doubled (resize_up2.hip), one batch of 3"""
    sf, df, dt = pair
    w, h = size
    rng = np.random.default_rng(w + h)
    frames = [(rng.random(w * h * 3 // 2) * (1023 if dt == np.uint16 else 255)).astype(dt) for _ in range(2)]
    srcs = [vali.Surface.Make(vali.PixelFormat[sf], w, h, gpu) for _ in range(3)]
    dsts = [vali.Surface.Make(vali.PixelFormat[df], w, h, gpu) for _ in range(3)]
    for i, s in enumerate(srcs):
        assert vali.PyFrameUploader(gpu).Run(frames[i % 2].view(np.uint8), s)[0]
    ud = vali.PySurfaceUD(gpu)
    assert ud.RunBatch(srcs, dsts) == (True, vali.TaskExecInfo.SUCCESS)
    def planar_want(host):
        off, want = 0, []
        for pw, ph in ((w, h), (w // 2, h // 2), (w // 2, h // 2)):
            plane = np.ascontiguousarray(host[off: off + pw * ph].reshape(ph, pw))
            want.append(oracle.resize_plane(plane, 1, w, h, "lanczos").reshape(-1))
            off += pw * ph
        return np.concatenate(want)
    wants = [planar_want(f) for f in frames]
    for i, d in enumerate(dsts):
        out = np.zeros(d.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(d, out)[0]
        assert np.array_equal(out.view(dt), np.asarray(wants[i % 2]).reshape(-1).view(dt))
    one = vali.Surface.Make(vali.PixelFormat[df], w, h, gpu)
    assert ud.Run(srcs[0], one) == (True, vali.TaskExecInfo.SUCCESS)
    out = np.zeros(one.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(one, out)[0]
    assert np.array_equal(out.view(dt), np.asarray(wants[0]).reshape(-1).view(dt))


@pytest.mark.parametrize("fmt,size", [("NV12", (1920, 1080)), ("RGB", (333, 77)), ("Y", (17, 5)), ("P10", (130, 66)),
                                      ("YUV420", (2050, 38)), ("RGB_PLANAR", (1366, 20)), ("YUV444_10bit", (1031, 9))])
@pytest.mark.parametrize("interp", ["lanczos", "cubic", "linear"])
def test_unchanged_size_is_a_copy(vali, gpu, oracle, fmt, size, interp):
    """every filter at 1:1 is the identity for integer element types (the specification's chains reduce to the centre
    tap): planes of unchanged size go through k_plane_copy -- rows that are not multiples of 16 bytes, several tiles wide"""
    w, h = size
    dt = DT.get(fmt, np.uint8)
    n = vali.Surface.Make(vali.PixelFormat[fmt], w, h, gpu).HostSize // np.dtype(dt).itemsize
    host = (np.random.default_rng(w + h).random(n) * (1023 if dt == np.uint16 else 255)).astype(dt)
    if fmt == "P10":
        host = (host.astype(np.uint16) << 6).astype(np.uint16)
    mode = {"lanczos": vali.Interpolation.LANCZOS, "cubic": vali.Interpolation.CUBIC, "linear": vali.Interpolation.LINEAR}[interp]
    got = roundtrip(vali, gpu, fmt, host, w, h, w, h, interp=mode)
    assert np.array_equal(got, host)
    assert np.array_equal(oracle.resize_surface(host, fmt, w, h, w, h, interp), host)   # (the specification agrees)
    with vali.tuning.Override(RESIZE_POINT=0):            # the arithmetic forms
        assert np.array_equal(roundtrip(vali, gpu, fmt, host, w, h, w, h, interp=mode), host)


# ---- exactly 3:2 both ways: the statically scheduled walk of resize_cols.hip (k_resize_cols_x32<..., SROWS>) --------------
@pytest.mark.parametrize("fmt", ["NV12", "YUV420", "Y", "P10", "YUV444", "RGB_PLANAR"])
@pytest.mark.parametrize("dst", [(16, 2), (16, 8), (32, 10), (496, 26), (1280, 720), (992, 100), (512, 258)])
def test_three_to_two_both_ways_bit_exact(vali, gpu, oracle, fmt, dst):
    dw, dh = dst
    if fmt in ("NV12", "YUV420", "P10"):
        dw, dh = 2 * dw, 2 * dh                         # so that the chroma planes are whole 8-element groups and row pairs too
    sw, sh = dw * 3 // 2, dh * 3 // 2
    dt = DT.get(fmt, np.uint8)
    n = vali.Surface.Make(vali.PixelFormat[fmt], sw, sh, gpu).HostSize // np.dtype(dt).itemsize
    rng = np.random.default_rng(dw * 3 + dh)
    host = (rng.random(n) * (1023 if dt == np.uint16 else 255)).astype(dt)
    host[:: max(1, n // 40)] = 1023 if dt == np.uint16 else 255
    if fmt == "P10":
        host = (host.astype(np.uint16) << 6).astype(np.uint16)
    want = oracle.resize_surface(host, fmt, sw, sh, dw, dh, "lanczos")
    for mode in (0, 1, 2, 3):                           # pairs per wave by launch size / 1 pair / the slot walk / 12 pairs
        with vali.tuning.Override(RESIZE_NO_SEPARABLE=mode):
            assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=vali.Interpolation.LANCZOS), want), mode


@pytest.mark.parametrize("geom", [(334, 78, 334, 78), (334, 78, 668, 156), (1366, 20, 1366, 20), (16, 8, 32, 16)])
@pytest.mark.parametrize("skew", [0, 1, 5])
def test_borrowed_surfaces_with_odd_pitch_and_base(vali, gpu, oracle, geom, skew):
    """Y planes borrowed from torch tensors (odd pitch, base off alignment by `skew` bytes, the last row ending where the
    buffer ends) through k_plane_copy (unchanged size) and k_resize_up2 (doubled): nothing outside the rows is read
    or written"""
    import torch

    sw, sh, dw, dh = geom
    rng = np.random.default_rng(sw + dh + skew)
    host = rng.integers(0, 256, sw * sh, dtype=np.uint8)
    sp, dpad = sw + 3, dw + 7
    sraw = torch.zeros(skew + (sh - 1) * sp + sw, dtype=torch.uint8, device="cuda")   # ends with the last row
    sview = torch.as_strided(sraw, (sh, sw), (sp, 1), skew)
    sview.copy_(torch.from_numpy(host.reshape(sh, sw)))
    draw = torch.full((skew + dh * dpad,), 0x5a, dtype=torch.uint8, device="cuda")
    dview = torch.as_strided(draw, (dh, dw), (dpad, 1), skew)
    torch.cuda.synchronize()
    src = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(sview), vali.Y)
    dst = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(dview), vali.Y)
    assert src.Pitch == sp and dst.Pitch == dpad
    rs = vali.PySurfaceResizer(vali.Y, gpu)                      # Lanczos, the reference's filter
    assert rs.Run(src, dst) == (True, vali.TaskExecInfo.SUCCESS)
    out = draw.cpu().numpy()
    got = np.lib.stride_tricks.as_strided(out[skew:], (dh, dw), (dpad, 1))
    want = oracle.resize_surface(host, "Y", sw, sh, dw, dh, "lanczos").reshape(dh, dw)
    assert np.array_equal(got, want)
    pad = np.lib.stride_tricks.as_strided(out[skew + dw:], (dh - 1, dpad - dw), (dpad, 1))
    assert np.all(pad == 0x5a) and np.all(out[:skew] == 0x5a)    # the bytes between and in front of the rows are untouched


# ---- planes that GROW vertically: filtered rows in registers (vali_amd/csrc/resize_rows.hip) -----------------------------
# geometry -> what it reaches: the benchmark's 720p -> 1080p; one tile per row (both image edges in the same tile); dst widths
# that are not multiples of 4 / of 2 (byte tails of the exchanged dwords); source widths that are not multiples of 4 (the
# last group patched); a single source row / column pair; > 3x enlargements (the emit loop behind the two straight-line
# rows); grow down the rows while the columns SHRINK a little (still fits 64 groups) or a lot (k_resize_taps keeps it);
# heights around the rows-per-wave forms
ROWS_GEOMS = [(1280, 720, 1920, 1080), (640, 360, 854, 480), (100, 40, 250, 90), (64, 48, 640, 480), (62, 30, 201, 67),
              (126, 50, 255, 129), (1918, 100, 2046, 110), (848, 464, 1278, 718), (2, 2, 14, 10), (4, 1, 9, 3),
              (300, 20, 310, 151), (1000, 300, 940, 420), (1200, 200, 400, 260), (130, 66, 516, 67)]


@pytest.mark.parametrize("fmt", ["NV12", "Y", "YUV420", "P10", "YUV444_10bit", "YUV422", "RGB_PLANAR", "RGB"])
@pytest.mark.parametrize("geom", ROWS_GEOMS)
@pytest.mark.parametrize("interp", ["lanczos", "cubic"])
def test_growing_planes_bit_exact(vali, gpu, oracle, fmt, geom, interp):
    sw, sh, dw, dh = geom
    if fmt in ("NV12", "YUV420", "P10") and ((sw | sh | dw | dh) & 1):
        pytest.skip("4:2:0 surfaces have even sizes")
    if fmt == "YUV422" and ((sw | dw) & 1):
        pytest.skip("4:2:2 surfaces have even widths")
    dt = DT.get(fmt, np.uint8)
    n = vali.Surface.Make(vali.PixelFormat[fmt], sw, sh, gpu).HostSize // np.dtype(dt).itemsize
    rng = np.random.default_rng(sw * 7 + dh)
    host = (rng.random(n) * (1023 if dt == np.uint16 else 255)).astype(dt)
    if dt == np.uint16:
        host[rng.integers(0, n, 20)] = 65535                     # saturation of the 16-bit store
    mode = vali.Interpolation.LANCZOS if interp == "lanczos" else vali.Interpolation.CUBIC
    want = oracle.resize_surface(host, fmt, sw, sh, dw, dh, interp)
    assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=mode), want)
    with vali.tuning.Override(RESIZE_ROWS=0):                    # round 2's kernel: the second implementation of the same bits
        assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=mode), want)
    with vali.tuning.Override(RESIZE_FORCE_GATHER=1):            # the direct-gather form inside the new kernel
        assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=mode), want)


# exactly 3:2 on both axes (k_resize_rows_x23): one lane / one tile / tile + a lane / source widths that are 2 (mod 4) (the
# last group moved left), halo groups that straddle the row's end, 1 .. 3 source row pairs, heights around the 12- and 48-row waves
X23_GEOMS = [(4, 2, 6, 3), (6, 4, 9, 6), (8, 6, 12, 9), (254, 34, 381, 51), (256, 36, 384, 54), (258, 10, 387, 15), (260, 12, 390, 18),
             (510, 20, 765, 30), (512, 8, 768, 12), (514, 6, 771, 9), (1282, 30, 1923, 45), (640, 360, 960, 540), (100, 66, 150, 99),
             (1280, 720, 1920, 1080), (128, 130, 192, 195), (516, 12, 774, 18), (268, 20, 402, 30), (1028, 24, 1542, 36), (8, 4, 12, 6)]


@pytest.mark.parametrize("fmt", ["NV12", "Y", "YUV420", "P10", "YUV444_10bit", "YUV444", "RGB"])
@pytest.mark.parametrize("geom", X23_GEOMS)
@pytest.mark.parametrize("interp", ["lanczos", "cubic"])
def test_three_to_two_enlargement_bit_exact(vali, gpu, oracle, fmt, geom, interp):
    sw, sh, dw, dh = geom
    if fmt in ("NV12", "YUV420", "P10") and ((sw | sh | dw | dh) & 1 or sw < 8):
        pytest.skip("4:2:0 surfaces have even sizes (and the chroma planes need 4 source pixels for this form)")
    dt = DT.get(fmt, np.uint8)
    n = vali.Surface.Make(vali.PixelFormat[fmt], sw, sh, gpu).HostSize // np.dtype(dt).itemsize
    rng = np.random.default_rng(sw * 3 + dh)
    host = (rng.random(n) * (1023 if dt == np.uint16 else 255)).astype(dt)
    host[rng.integers(0, n, max(4, n // 50))] = 0                # runs of zeros: the sign of zero in the shifted-weight chains
    if dt == np.uint16:
        host[rng.integers(0, n, 20)] = 65535
    mode = vali.Interpolation.LANCZOS if interp == "lanczos" else vali.Interpolation.CUBIC
    want = oracle.resize_surface(host, fmt, sw, sh, dw, dh, interp)
    assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=mode), want)
    for rows_mode in (1, 3):                                     # 12-row and 48-row waves
        with vali.tuning.Override(RESIZE_NO_SEPARABLE=rows_mode):
            assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=mode), want), rows_mode
    with vali.tuning.Override(RESIZE_ROWS=2):                    # the general growing-planes kernel on the same geometry
        assert np.array_equal(roundtrip(vali, gpu, fmt, host, sw, sh, dw, dh, interp=mode), want)


def test_three_to_two_enlargement_all_zero_and_flat_frames(vali, gpu, oracle):
    """flat frames: every chain of the shifted-weight form sees zeros / equal values in every slot"""
    for val in (0, 255, 16):
        host = np.full(260 * 12 * 3 // 2, val, np.uint8)
        want = oracle.resize_surface(host, "NV12", 260, 12, 390, 18, "lanczos")
        assert np.array_equal(roundtrip(vali, gpu, "NV12", host, 260, 12, 390, 18, interp=vali.Interpolation.LANCZOS), want), val


@pytest.mark.parametrize("rows_mode", [0, 1, 2, 3])
def test_growing_planes_batch_and_rows_per_wave(vali, gpu, oracle, rows_mode):
    """a batch under every rows-per-wave form (2 / 8 / 32), NV12 (both channel counts in one launch) and P10"""
    for fmt, dt, top in (("NV12", np.uint8, 256), ("P10", np.uint16, 1024)):
        sw, sh, dw, dh, n = 640, 360, 962, 542, 5
        rng = np.random.default_rng(91)
        pf = vali.PixelFormat[fmt]
        frames = [rng.integers(0, top, sw * sh * 3 // 2).astype(dt) for _ in range(2)]
        srcs = [vali.Surface.Make(pf, sw, sh, gpu) for _ in range(n)]
        dsts = [vali.Surface.Make(pf, dw, dh, gpu) for _ in range(n)]
        for i, s_ in enumerate(srcs):
            assert vali.PyFrameUploader(gpu).Run(frames[i % 2].view(np.uint8), s_)[0]
        with vali.tuning.Override(RESIZE_NO_SEPARABLE=rows_mode):
            assert vali.PySurfaceResizer(pf, gpu).RunBatch(srcs, dsts) == (True, vali.TaskExecInfo.SUCCESS)
        wants = [oracle.resize_surface(f, fmt, sw, sh, dw, dh, "lanczos") for f in frames]
        for i, d in enumerate(dsts):
            out = np.zeros(d.HostSize, np.uint8)
            assert vali.PySurfaceDownloader(gpu).Run(d, out)[0]
            assert np.array_equal(out.view(dt), wants[i % 2]), (fmt, i)


@pytest.mark.parametrize("geom", [(334, 78, 501, 117), (333, 40, 500, 61), (16, 8, 40, 17)])
@pytest.mark.parametrize("skew", [0, 1, 5])
def test_growing_planes_borrowed_surfaces_with_tight_pitch(vali, gpu, oracle, geom, skew):
    """Y planes borrowed from torch tensors whose pitch IS the width (+0 / +3) and whose last row ends where the buffer
    ends: a width that is not a multiple of 4 with no padding behind it takes the direct-gather form (the 4-pixel groups
    of the staged form would read past the buffer), padded ones the staged form with misaligned rows"""
    import torch

    sw, sh, dw, dh = geom
    for extra in (0, 3):
        rng = np.random.default_rng(sw + dh + skew + extra)
        host = rng.integers(0, 256, sw * sh, dtype=np.uint8)
        sp, dpad = sw + extra, dw + 7
        sraw = torch.zeros(skew + (sh - 1) * sp + sw, dtype=torch.uint8, device="cuda")   # ends with the last row
        sview = torch.as_strided(sraw, (sh, sw), (sp, 1), skew)
        sview.copy_(torch.from_numpy(host.reshape(sh, sw)))
        draw = torch.full((skew + dh * dpad,), 0x5a, dtype=torch.uint8, device="cuda")
        dview = torch.as_strided(draw, (dh, dw), (dpad, 1), skew)
        torch.cuda.synchronize()
        src = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(sview), vali.Y)
        dst = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(dview), vali.Y)
        assert vali.PySurfaceResizer(vali.Y, gpu).Run(src, dst) == (True, vali.TaskExecInfo.SUCCESS)
        out = draw.cpu().numpy()
        got = np.lib.stride_tricks.as_strided(out[skew:], (dh, dw), (dpad, 1))
        want = oracle.resize_surface(host, "Y", sw, sh, dw, dh, "lanczos").reshape(dh, dw)
        assert np.array_equal(got, want), extra
        pad = np.lib.stride_tricks.as_strided(out[skew + dw:], (dh - 1, dpad - dw), (dpad, 1))
        assert np.all(pad == 0x5a) and np.all(out[:skew] == 0x5a)


@pytest.mark.parametrize("geom", [(334, 78, 501, 117), (8, 4, 12, 6), (130, 6, 195, 9), (258, 50, 387, 75)])
@pytest.mark.parametrize("skew", [0, 1, 6])
def test_three_to_two_enlargement_of_borrowed_packed_rgb(vali, gpu, oracle, geom, skew):
    """packed RGB enlarged 3:2 (a lane owns 2 pixels = 6 bytes at ANY even offset and stores 9 at any offset): surfaces
    borrowed from torch whose pitch is the row (+0 / +5 bytes), whose base is skewed and whose last row ends where the
    buffer ends -- nothing may be read past it, nothing written beside the rows"""
    import torch

    sw, sh, dw, dh = geom
    for extra in (0, 5):
        rng = np.random.default_rng(sw + dh + skew + extra)
        host = rng.integers(0, 256, sw * sh * 3, dtype=np.uint8)
        sp, dpad = sw * 3 + extra, dw * 3 + 7
        sraw = torch.zeros(skew + (sh - 1) * sp + sw * 3, dtype=torch.uint8, device="cuda")   # ends with the last row
        sview = torch.as_strided(sraw, (sh, sw * 3), (sp, 1), skew)
        sview.copy_(torch.from_numpy(host.reshape(sh, sw * 3)))
        draw = torch.full((skew + dh * dpad,), 0x5a, dtype=torch.uint8, device="cuda")
        dview = torch.as_strided(draw, (dh, dw * 3), (dpad, 1), skew)
        torch.cuda.synchronize()
        src = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(sview), vali.RGB)
        dst = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(dview), vali.RGB)
        assert (src.Width, src.Height, dst.Width, dst.Height) == (sw, sh, dw, dh)
        assert vali.PySurfaceResizer(vali.RGB, gpu).Run(src, dst) == (True, vali.TaskExecInfo.SUCCESS)
        out = draw.cpu().numpy()
        got = np.lib.stride_tricks.as_strided(out[skew:], (dh, dw * 3), (dpad, 1))
        want = oracle.resize_surface(host, "RGB", sw, sh, dw, dh, "lanczos").reshape(dh, dw * 3)
        assert np.array_equal(got, want), extra
        pad = np.lib.stride_tricks.as_strided(out[skew + dw * 3:], (dh - 1, dpad - dw * 3), (dpad, 1))
        assert np.all(pad == 0x5a) and np.all(out[:skew] == 0x5a)


@pytest.mark.parametrize("geom", [(300, 40, 375, 51), (16, 8, 40, 17), (130, 30, 259, 61), (1000, 12, 1003, 13)])
@pytest.mark.parametrize("skew", [0, 1, 6])
def test_growing_packed_rgb_borrowed_surfaces_with_tight_pitch(vali, gpu, oracle, geom, skew):
    """packed RGB enlarged at general ratios (k_resize_rows_rgb: a lane loads 24 bytes from its own byte address; tiles at
    the image's edges load their pixels one by one, 4 bytes that end with the pixel): borrowed surfaces whose pitch is the
    row (+0 / +5 bytes), whose base is skewed and whose last row ends where the buffer ends; one tile per row, several, a
    ragged last tile, an odd last pixel"""
    import torch

    sw, sh, dw, dh = geom
    for extra in (0, 5):
        rng = np.random.default_rng(sw + dh + skew + extra)
        host = rng.integers(0, 256, sw * sh * 3, dtype=np.uint8)
        sp, dpad = sw * 3 + extra, dw * 3 + 7
        sraw = torch.zeros(skew + (sh - 1) * sp + sw * 3, dtype=torch.uint8, device="cuda")   # ends with the last row
        sview = torch.as_strided(sraw, (sh, sw * 3), (sp, 1), skew)
        sview.copy_(torch.from_numpy(host.reshape(sh, sw * 3)))
        draw = torch.full((skew + dh * dpad,), 0x5a, dtype=torch.uint8, device="cuda")
        dview = torch.as_strided(draw, (dh, dw * 3), (dpad, 1), skew)
        torch.cuda.synchronize()
        src = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(sview), vali.RGB)
        dst = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(dview), vali.RGB)
        assert vali.PySurfaceResizer(vali.RGB, gpu).Run(src, dst) == (True, vali.TaskExecInfo.SUCCESS)
        out = draw.cpu().numpy()
        got = np.lib.stride_tricks.as_strided(out[skew:], (dh, dw * 3), (dpad, 1))
        want = oracle.resize_surface(host, "RGB", sw, sh, dw, dh, "lanczos").reshape(dh, dw * 3)
        assert np.array_equal(got, want), extra
        pad = np.lib.stride_tricks.as_strided(out[skew + dw * 3:], (dh - 1, dpad - dw * 3), (dpad, 1))
        assert np.all(pad == 0x5a) and np.all(out[:skew] == 0x5a)


def test_tap_tables_across_streams_and_inside_a_capture(vali, gpu, oracle):
    """The columns-first kernels of general ratios read their taps from per-geometry tables written by the first call
    (vali_amd/csrc/tap_table.hip).  A geometry nobody has used yet: (a) its first call on one stream and, right behind
    it, a call on ANOTHER stream (which has to wait for the writer) both give the oracle's bits; (b) recorded into a
    graph before any table exists for it, the kernels compute their taps themselves -- nothing is allocated or
    launched for a table during a capture -- and every replay gives the oracle's bits."""
    from conftest import make_nv12

    shim = vali._native.shim
    sw, sh = 1280, 720
    nv = make_nv12(sw, sh, 11)
    s1, s2 = shim.stream_create(gpu), shim.stream_create(gpu)
    src = vali.Surface.Make(vali.NV12, sw, sh, gpu)
    assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), src)[0]

    def check(dst, dw, dh):
        out = np.zeros(dst.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(dst, out)[0]
        assert np.array_equal(out, oracle.resize_surface(nv.reshape(-1), "NV12", sw, sh, dw, dh, "lanczos"))

    # (a) two streams, a geometry of its own (sizes no other test uses)
    dw, dh = 846, 474
    r1 = vali.PySurfaceResizer(vali.NV12, gpu, s1, interpolation=vali.Interpolation.LANCZOS)
    r2 = vali.PySurfaceResizer(vali.NV12, gpu, s2, interpolation=vali.Interpolation.LANCZOS)
    d1, d2 = vali.Surface.Make(vali.NV12, dw, dh, gpu), vali.Surface.Make(vali.NV12, dw, dh, gpu)
    assert r1.RunAsync(src, d1)[0] and r2.RunAsync(src, d2)[0]
    shim.stream_sync(gpu, s1)
    shim.stream_sync(gpu, s2)
    check(d1, dw, dh)
    check(d2, dw, dh)
    # (b) captured before its tables exist (the kernel itself was loaded by (a): same format, same slots)
    dw, dh = 842, 470
    d3 = vali.Surface.Make(vali.NV12, dw, dh, gpu)
    cap = vali.StreamCapture(s1, gpu).Keep(src, d3)
    with cap:
        assert r1.RunAsync(src, d3)[0]
    for _ in range(2):
        shim.memset2d_async(gpu, d3._planes[0].GpuMem, d3._planes[0].Pitch, 0, dw, dh * 3 // 2, s1)
        cap.Launch()
        shim.stream_sync(gpu, s1)
        check(d3, dw, dh)
    # ... and afterwards eagerly, now through a table
    assert r1.Run(src, d3)[0]
    check(d3, dw, dh)
