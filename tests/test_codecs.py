"""Decoder / encoder stand-ins (vali_amd/codecs.py): CPU-mode raw reader needs no GPU."""
import numpy as np
import pytest

from conftest import GOLDEN


def test_raw_decoder_cpu_mode(vali):
    dec = vali.PyDecoder(str(GOLDEN / "test_small_2frames.nv12"), {"video_size": "424x232"}, gpu_id=-1)
    assert (dec.Width, dec.Height, dec.Format, dec.NumFrames) == (424, 232, vali.NV12, 2)
    assert not dec.IsAccelerated and dec.HostFrameSize == 424 * 232 * 3 // 2
    raw = np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8).reshape(2, -1)
    frame = np.ndarray(shape=(0,), dtype=np.uint8)
    for i in range(2):
        ok, info = dec.DecodeSingleFrame(frame)
        assert ok and info == vali.TaskExecInfo.SUCCESS and np.array_equal(frame, raw[i])
    assert dec.DecodeSingleFrame(frame) == (False, vali.TaskExecInfo.END_OF_STREAM)
    # surface decode is refused in CPU mode (PyDecoder.cpp:98-101)
    assert dec.DecodeSingleSurface(None)[0] is False


def test_compressed_input_and_encoders_raise(vali, tmp_path):
    with pytest.raises(RuntimeError):
        vali.PyDecoder(str(tmp_path / "movie.mp4"), {}, gpu_id=-1)
    if not vali.codecs.have_av():
        with pytest.raises(RuntimeError, match="PyAV"):      # the only condition under which the encoder class raises
            vali.PyNvEncoder({"s": "64x48"}, 0)
    with pytest.raises(ValueError):                              # TaskNvJpegEncode.cpp:123
        vali.NvJpegEncodeContext(90, vali.NV12)
    ctx = vali.NvJpegEncodeContext(90, vali.RGB)
    assert ctx.Compression() == 90 and ctx.Format() == vali.RGB


@pytest.mark.gpu
def test_decode_convert_download_pipeline(vali, gpu, oracle):
    """The canonical user pipeline (samples/sample_decode_show.ipynb): decode -> NV12->RGB ->
    download, on the decoder's stream, here fed from raw NV12."""
    dec = vali.PyDecoder(str(GOLDEN / "test_small_2frames.nv12"),
                         {"f": "rawvideo", "video_size": "424x232", "pixel_format": "nv12"}, gpu_id=gpu)
    cvt = vali.PySurfaceConverter(gpu, dec.Stream)
    dwn = vali.PySurfaceDownloader(gpu, dec.Stream)
    surf = vali.Surface.Make(dec.Format, dec.Width, dec.Height, gpu)
    rgb = vali.Surface.Make(vali.RGB, dec.Width, dec.Height, gpu)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    raw = np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8).reshape(2, -1)
    n = 0
    while True:
        ok, info = dec.DecodeSingleSurface(surf)
        if not ok:
            assert info == vali.TaskExecInfo.END_OF_STREAM
            break
        assert cvt.Run(surf, rgb, cc)[0]
        out = np.zeros(rgb.HostSize, np.uint8)
        assert dwn.Run(rgb, out)[0]
        want = oracle.convert(raw[n], "NV12", "RGB", 424, 232, oracle.cvt_params(csc_variant=1))
        assert np.array_equal(out, want)
        n += 1
    assert n == 2
    assert dec.DecodeSingleFrame(np.zeros(1, np.uint8)) == (False, vali.TaskExecInfo.FAIL)


@pytest.mark.gpu
def test_frame_converter_runs_on_the_gpu(vali, gpu, oracle):
    """reference tests/test_PyFrameConverter.py:59-102 (NV12 -> RGB on ndarrays)."""
    w, h = 424, 232
    raw = np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8).reshape(2, -1)
    fc = vali.PyFrameConverter(w, h, vali.NV12, vali.RGB)
    dst = np.ndarray(shape=(0,), dtype=np.uint8)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    ok, info = fc.Run(raw[0], dst, cc)
    assert ok and dst.size == w * h * 3 and fc.Format == vali.RGB
    assert np.array_equal(dst, oracle.convert(raw[0], "NV12", "RGB", w, h, oracle.cvt_params(csc_variant=1)))
    assert fc.Run(raw[0][:-1], dst, cc) == (False, vali.TaskExecInfo.INVALID_INPUT)


# ---- compressed input through PyAV (optional) -------------------------------------------------------------
class _FakeFrame:
    def __init__(self, yuv420, w, h, pts=0, key=False):
        self._p, self._w, self._h, self.pts, self.key_frame = yuv420, w, h, pts, key

    def to_ndarray(self, format):  # noqa: A002 -- PyAV's keyword
        y, c = self._w * self._h, self._w * self._h // 4
        if format == "yuv420p":
            return self._p.reshape(self._h * 3 // 2, self._w)
        assert format == "nv12"
        uv = np.empty(2 * c, np.uint8)
        uv[0::2], uv[1::2] = self._p[y:y + c], self._p[y + c:]
        return np.concatenate([self._p[:y], uv]).reshape(self._h * 3 // 2, self._w)


def _fake_av(frames, w, h):
    """A stand-in for the `av` module with the handful of attributes _AvSource touches, so the adapter logic
    (format choice per mode, properties, end of stream) is exercised where PyAV is not installed."""
    import types
    from fractions import Fraction

    # 30 fps in a 1/15360 time base (512 ticks per frame), the stream starts at tick 1024, a key frame every 4 frames
    cc = types.SimpleNamespace(width=w, height=h, pix_fmt="yuv420p", colorspace=1, color_range=1, gop_size=4, bit_rate=123456)
    stream = types.SimpleNamespace(codec_context=cc, average_rate=Fraction(30, 1), guessed_rate=Fraction(30, 1), frames=len(frames),
                                   format=types.SimpleNamespace(name="yuv420p"), time_base=Fraction(1, 15360), start_time=1024,
                                   duration=512 * len(frames), index=0, metadata={"handler_name": "VideoHandler"})
    state = {"at": 0, "seeks": []}

    def decode(s):
        def gen():
            while state["at"] < len(frames):
                i = state["at"]
                state["at"] += 1
                yield _FakeFrame(frames[i], w, h, pts=1024 + 512 * i, key=(i % 4 == 0))
        return gen()

    def seek(offset, stream=None, backward=True, any_frame=False):      # lands on the key frame at or before `offset`
        state["seeks"].append((offset, backward, any_frame))
        i = max(0, min(len(frames) - 1, (offset - 1024) // 512))
        state["at"] = i - i % 4
    container = types.SimpleNamespace(streams=types.SimpleNamespace(video=[stream]), close=lambda: None, decode=decode, seek=seek,
                                      metadata={"title": "clip"})
    return types.SimpleNamespace(open=lambda path, options=None: container, _state=state, _stream=stream)


def test_compressed_input_through_pyav_adapter_cpu_mode(vali, monkeypatch, tmp_path):
    import sys
    w, h = 64, 48
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8) for _ in range(3)]
    monkeypatch.setitem(sys.modules, "av", _fake_av(frames, w, h))
    dec = vali.PyDecoder(str(tmp_path / "movie.mp4"), {}, gpu_id=-1)
    assert (dec.Width, dec.Height, dec.Format, dec.NumFrames, dec.Framerate) == (w, h, vali.YUV420, 3, 30.0)
    assert dec.ColorSpace == vali.ColorSpace.BT_709 and dec.ColorRange == vali.ColorRange.MPEG and not dec.IsAccelerated
    out = np.ndarray(shape=(0,), dtype=np.uint8)
    for f in frames:
        assert dec.DecodeSingleFrame(out) == (True, vali.TaskExecInfo.SUCCESS) and np.array_equal(out, f)
    assert dec.DecodeSingleFrame(out) == (False, vali.TaskExecInfo.END_OF_STREAM)


@pytest.mark.gpu
def test_compressed_input_through_pyav_adapter_uploads_nv12(vali, gpu, monkeypatch, tmp_path):
    import sys
    w, h = 64, 48
    rng = np.random.default_rng(1)
    frames = [rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8) for _ in range(2)]
    monkeypatch.setitem(sys.modules, "av", _fake_av(frames, w, h))
    dec = vali.PyDecoder(str(tmp_path / "movie.mkv"), {}, gpu_id=gpu)
    assert dec.Format == vali.NV12 and dec.IsAccelerated
    surf = vali.Surface.Make(vali.NV12, w, h, gpu)
    for f in frames:
        assert dec.DecodeSingleSurface(surf) == (True, vali.TaskExecInfo.SUCCESS)
        got = np.zeros(surf.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu, dec.Stream).Run(surf, got)[0]
        assert np.array_equal(got, _FakeFrame(f, w, h).to_ndarray("nv12").reshape(-1))
    assert dec.DecodeSingleSurface(surf) == (False, vali.TaskExecInfo.END_OF_STREAM)


def test_seek_through_the_pyav_adapter_decodes_forward_to_the_target(vali, monkeypatch, tmp_path):
    """reference TaskDecodeFrame.cpp:944-1029 (SeekDecode): seek backward to the key frame, decode forward to the frame whose
    pts reaches the target; the target includes the stream's start_time; VFR + seek by number is NOT_SUPPORTED."""
    import sys
    from fractions import Fraction
    w, h = 32, 16
    frames = [np.full(w * h * 3 // 2, i, np.uint8) for i in range(12)]
    fake = _fake_av(frames, w, h)
    monkeypatch.setitem(sys.modules, "av", fake)
    dec = vali.PyDecoder(str(tmp_path / "movie.mp4"), {}, gpu_id=-1)
    out, pd = np.ndarray(shape=(0,), dtype=np.uint8), vali.PacketData()
    # frame 6 is not a key frame: the container lands on 4, frames 4 and 5 are decoded and dropped
    assert dec.DecodeSingleFrame(out, pd, vali.SeekContext(6)) == (True, vali.TaskExecInfo.SUCCESS)
    assert out[0] == 6 and pd.pts == 1024 + 512 * 6 and pd.key == 0
    assert fake._state["seeks"][-1] == (1024 + 512 * 6, True, False)          # time-base units, start_time included, backward
    assert dec.DecodeSingleFrame(out, pd)[0] and out[0] == 7                    # decoding continues behind the target
    assert dec.DecodeSingleFrame(out, pd, vali.SeekContext(0.1))[0] and out[0] == 3      # 0.1 s = tick 1536 -> frame 3
    assert dec.DecodeSingleFrame(out, pd, vali.SeekContext(8))[0] and out[0] == 8 and pd.key == 1
    # KEY_FRAMES mode: one decode after the seek, i.e. the key frame itself
    assert dec.Mode == vali.DecodeMode.ALL_FRAMES
    dec.SetMode(vali.DecodeMode.KEY_FRAMES)
    assert dec.Mode == vali.DecodeMode.KEY_FRAMES
    assert dec.DecodeSingleFrame(out, pd, vali.SeekContext(6))[0] and out[0] == 4
    dec.SetMode(vali.DecodeMode.ALL_FRAMES)
    with pytest.raises(TypeError):
        dec.SetMode(1)
    # past the end: END_OF_STREAM, not a stale frame
    assert dec.DecodeSingleFrame(out, pd, vali.SeekContext(400)) == (False, vali.TaskExecInfo.END_OF_STREAM)
    # the read-only ancillaries (PyDecoder.cpp:563-680)
    assert (dec.GopSize, dec.Bitrate, dec.NumStreams, dec.StreamIndex) == (4, 123456, 1, 0)
    assert abs(dec.Timebase - 1 / 15360) < 1e-12 and abs(dec.StartTime - 1024 / 15360) < 1e-12 and abs(dec.Duration - 0.4) < 1e-9
    assert dec.Metadata == {"context": {"title": "clip"}, "video_stream": {"handler_name": "VideoHandler"}}
    assert dec.MotionVectors == [] and not dec.IsVFR and dec.Framerate == 30.0 and dec.AvgFramerate == 30.0
    # variable frame rate (r_frame_rate != avg_frame_rate): by number refused, by timestamp served
    fake._stream.average_rate = Fraction(2997, 100)
    vfr = vali.PyDecoder(str(tmp_path / "vfr.mp4"), {}, gpu_id=-1)
    assert vfr.IsVFR
    assert vfr.DecodeSingleFrame(out, pd, vali.SeekContext(6)) == (False, vali.TaskExecInfo.NOT_SUPPORTED)
    assert vfr.DecodeSingleFrame(out, pd, vali.SeekContext(0.2))[0] and out[0] == 6


def test_log_level_and_probe_validate_their_arguments(vali, tmp_path):
    """reference VALI.cpp:206-214, 512-521; PyDecoder.cpp:684-700."""
    assert [int(v) for v in (vali.FfmpegLogLevel.PANIC, vali.FfmpegLogLevel.ERROR, vali.FfmpegLogLevel.DEBUG)] == [0, 16, 48]
    assert vali.ERROR == vali.FfmpegLogLevel.ERROR and vali.KEY_FRAMES == vali.DecodeMode.KEY_FRAMES     # export_values
    vali.SetFFMpegLogLevel(vali.FfmpegLogLevel.ERROR)
    with pytest.raises(TypeError):
        vali.SetFFMpegLogLevel(16)
    sp = vali.StreamParams()
    assert (sp.width, sp.num_frames, sp.fps, sp.color_space) == (0, 0, 0.0, vali.ColorSpace.UNSPEC)
    from vali_amd.codecs import have_av
    if not have_av():
        with pytest.raises(RuntimeError, match="PyAV"):
            vali.PyDecoder.Probe(str(tmp_path / "x.mp4"))
    # raw video answers the ancillaries from the file itself
    w, h, n = 32, 16, 5
    path = tmp_path / "clip.nv12"
    np.zeros(n * w * h * 3 // 2, np.uint8).tofile(path)
    dec = vali.PyDecoder(str(path), {"video_size": f"{w}x{h}", "framerate": "10"}, gpu_id=-1)
    assert (dec.GopSize, dec.NumStreams, dec.Duration, dec.Timebase, dec.Metadata) == (1, 1, 0.5, 0.1, {})
    assert dec.Bitrate == w * h * 3 // 2 * 8 * 10 and dec.Mode == vali.DecodeMode.ALL_FRAMES


def test_real_pyav_decodes_the_reference_video_when_installed(vali):
    """With a real PyAV and the reference's data next to it (never the case in the build container): frame 0 of
    test.mp4, resized 2:1, is the committed fixture test_small.nv12."""
    av = pytest.importorskip("av")  # noqa: F841
    import os
    path = os.environ.get("VALI_REFERENCE_VIDEO", "/root/reference/tests/data/test.mp4")
    if not os.path.exists(path):
        pytest.skip("reference video not present")
    dec = vali.PyDecoder(path, {}, gpu_id=-1)
    assert (dec.Width, dec.Height, dec.Format) == (848, 464, vali.YUV420)
    frame = np.ndarray(shape=(0,), dtype=np.uint8)
    assert dec.DecodeSingleFrame(frame)[0]
    small = np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8)[:424 * 232].reshape(232, 424)
    assert np.array_equal(frame[:848 * 464].reshape(464, 848)[0::2, 0::2], small)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["RGB", "YUV420", "RGB_PLANAR", "YUV444"])
def test_jpeg_encoder_cpu_fallback(vali, gpu, oracle, fmt):
    """reference tests/test_PyNvJpegEncoder.py:150-222: NV12 -> dst format on the GPU -> JPEG; for RGB the decoded
    image must be within 42 dB of the raw surface.  Here the compression runs on the CPU (Pillow)."""
    PIL = pytest.importorskip("PIL.Image")
    import io

    w, h = 424, 232
    raw = np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8).reshape(2, -1)
    src = vali.Surface.Make(vali.NV12, w, h, gpu)
    assert vali.PyFrameUploader(gpu).Run(raw[0], src)[0]
    dst_fmt = vali.PixelFormat[fmt]
    cvt = vali.PySurfaceConverter(gpu)
    if fmt in ("RGB", "YUV420"):
        dst = vali.Surface.Make(dst_fmt, w, h, gpu)
        assert cvt.Run(src, dst)[0]
    else:                                            # two steps, as a reference user would chain them
        mid = vali.Surface.Make(vali.RGB, w, h, gpu)
        dst = vali.Surface.Make(dst_fmt, w, h, gpu)
        assert cvt.Run(src, mid)[0] and cvt.Run(mid, dst)[0]
    enc = vali.PyNvJpegEncoder(gpu_id=gpu)
    ctx = enc.Context(compression=100, pixel_format=dst_fmt)
    buffers, info = enc.Run(ctx, [dst, dst])
    assert info == vali.TaskExecInfo.SUCCESS and len(buffers) == 2 and buffers[0].dtype == np.uint8 and buffers[0].size > 1000
    img = PIL.open(io.BytesIO(buffers[0].tobytes()))
    assert img.size == (w, h)
    if fmt == "RGB":
        host = np.zeros(dst.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(dst, host)[0]
        d = np.asarray(img).astype(np.float64).reshape(-1) - host
        assert 10 * np.log10(255.0 ** 2 / np.mean(d * d)) >= 42.0
    # all or nothing: a surface of another format fails the whole call (PyNvJpegEncoder.cpp:66-69)
    other = vali.Surface.Make(vali.BGR if fmt != "BGR" else vali.RGB, w, h, gpu)
    assert enc.Run(ctx, [dst, other]) == ([], vali.TaskExecInfo.FAIL)


# ---- PacketData / SeekContext (VALI.cpp:216-279) and the encoder over PyAV -------------------------------------------
def test_packet_data_and_seek_context_exist_and_drive_the_raw_decoder(vali, tmp_path):
    pd = vali.PacketData()
    assert (pd.key, pd.pts, pd.dts, pd.pos, pd.bsl, pd.duration) == (0, 0, 0, 0, 0, 0) and "pts:" in repr(pd)
    by_frame, by_ts = vali.SeekContext(3), vali.SeekContext(0.2)
    assert (by_frame.seek_frame, by_ts.seek_tssec) == (3, 0.2) and vali.SeekContext(seek_ts=1.5).seek_tssec == 1.5
    with pytest.raises(TypeError):
        vali.SeekContext()
    w, h, n = 32, 16, 6
    frames = np.arange(n * w * h * 3 // 2, dtype=np.uint32).astype(np.uint8).reshape(n, -1)
    path = tmp_path / "clip.yuv"
    frames.tofile(path)
    dec = vali.PyDecoder(str(path), {"video_size": f"{w}x{h}", "framerate": "10"}, gpu_id=-1)
    out = np.ndarray(shape=(0,), dtype=np.uint8)
    assert dec.DecodeSingleFrame(out, pd) == (True, vali.TaskExecInfo.SUCCESS) and pd.pts == 0 and pd.key == 1
    assert dec.DecodeSingleFrame(out, pd, by_frame)[0] and np.array_equal(out, frames[3]) and pd.pts == 3   # reference-style loop
    assert dec.DecodeSingleFrame(out, pd, by_ts)[0] and np.array_equal(out, frames[2])                     # 0.2 s at 10 fps
    assert dec.DecodeSingleFrame(out, pd)[0] and np.array_equal(out, frames[3])


class _FakeEncoder:
    """the handful of av.CodecContext attributes PyNvEncoder touches; a 'packet' is the frame's first 8 bytes, one frame late"""

    def __init__(self, delay=1):
        self.options, self._held, self._eof, self._delay = {}, None, False, delay

    def encode(self, frame):
        if frame is not None and self._eof:
            raise EOFError("avcodec_send_frame: AVERROR_EOF")      # what libavcodec does to a drained encoder
        if frame is None:
            self._eof = True
        if self._delay == 0 and frame is not None:
            return [bytes(frame.data[:8])]
        out = [] if self._held is None else [self._held]
        self._held = None if frame is None else bytes(frame.data[:8])
        return out


@pytest.mark.gpu
def test_video_encoder_through_the_pyav_adapter(vali, gpu, monkeypatch):
    import sys
    import types
    made = []

    def create(name, mode):
        assert mode == "w"
        if name == "libx264":
            raise ValueError("no such encoder")        # falls through to the next name
        made.append(name)
        return _FakeEncoder()
    frame_cls = types.SimpleNamespace(from_ndarray=lambda a, format: types.SimpleNamespace(data=a.reshape(-1).tobytes(), pts=0))
    monkeypatch.setitem(sys.modules, "av", types.SimpleNamespace(CodecContext=types.SimpleNamespace(create=create), VideoFrame=frame_cls))
    w, h = 64, 48
    enc = vali.PyNvEncoder({"s": f"{w}x{h}", "codec": "h264", "fps": "25", "bitrate": "2M", "gop": "30", "preset": "fast"}, gpu)
    assert made == ["h264"] and (enc.Width, enc.Height, enc.Format) == (w, h, vali.NV12) and enc.FrameSizeInBytes == w * h * 3 // 2
    surf = vali.Surface.Make(vali.NV12, w, h, gpu)
    rng = np.random.default_rng(3)
    pkt = np.ndarray(shape=(0,), dtype=np.uint8)
    sent = []
    for i in range(3):
        nv12 = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
        assert vali.PyFrameUploader(gpu).Run(nv12, surf)[0]
        sent.append(nv12[:8].copy())
        got = enc.EncodeSingleSurface(surf, pkt)
        assert got == (i > 0)                           # the encoder holds one frame back: False is "nothing yet", not an error
        if got:
            assert np.array_equal(pkt, sent[i - 1])     # luma bytes survive the NV12 -> planar repack in front of the codec
    assert enc.Flush(pkt) and np.array_equal(pkt, sent[2]) and not enc.Flush(pkt)
    # a drained libavcodec encoder is at EOF: the next frame gets a NEW context instead of an EOFError
    assert not enc.EncodeSingleSurface(surf, pkt) and made == ["h264", "h264"]
    # sync mode never drains per frame: two frames in a row, the second returns the first one's packet
    assert enc.EncodeSingleSurface(surf, pkt, sync=True) and np.array_equal(pkt, sent[2])
    assert enc._ctx.options.get("preset") == "fast" and enc._ctx.max_b_frames == 0 and enc._ctx.thread_count == 1
    assert not enc.EncodeSingleSurface(vali.Surface.Make(vali.NV12, 32, 32, gpu), pkt)      # wrong size
    with pytest.raises(RuntimeError):
        vali.PyNvEncoder({"codec": "h264"}, gpu)        # no size


def test_pyav_frames_are_repacked_from_their_planes(vali):
    """_frame_to_flat builds nv12 / p010le / planar layouts from the PLANES of a planar frame (line_size > width honoured): no
    reliance on to_ndarray(format='p010le'), which several PyAV releases reject"""
    import types
    from vali_amd.codecs import _frame_to_flat
    w, h, ls = 16, 8, 24

    def frame(dt, name):
        rng = np.random.default_rng(9)
        planes, packed = [], []
        for pw, ph in ((w, h), (w // 2, h // 2), (w // 2, h // 2)):
            a = rng.integers(0, 1024 if dt == np.uint16 else 256, (ph, pw)).astype(dt)
            pad = np.zeros((ph, ls // (2 if pw < w else 1) * dt().itemsize), np.uint8)
            pad[:, :pw * dt().itemsize] = a.view(np.uint8).reshape(ph, -1)
            planes.append(pad.tobytes())
            packed.append(a.reshape(-1))
        f = types.SimpleNamespace(planes=planes, format=types.SimpleNamespace(name=name))
        f.reformat = lambda format: f
        return f, packed
    f8, (y, u, v) = frame(np.uint8, "yuv420p")
    assert np.array_equal(_frame_to_flat(f8, "yuv420p", w, h), np.concatenate([y, u, v]))
    uv = np.empty(u.size * 2, np.uint8); uv[0::2], uv[1::2] = u, v
    assert np.array_equal(_frame_to_flat(f8, "nv12", w, h), np.concatenate([y, uv]))
    f10, (y, u, v) = frame(np.uint16, "yuv420p10le")
    uv = np.empty(u.size * 2, np.uint16); uv[0::2], uv[1::2] = u, v
    assert np.array_equal(_frame_to_flat(f10, "p010le", w, h).view(np.uint16), np.concatenate([y, uv]) << 6)
