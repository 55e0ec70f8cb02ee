"""Callers on either side of the surface path: decoder / encoder / CPU-converter classes.

The reference implements these with FFmpeg + NVDEC/NVENC/nvJPEG (SURVEY.md section 2 rows
12-15): decode/encode ASICs and a demuxer are outside this build's scope and no FFmpeg exists
offline.  What IS provided so pipelines written against python_vali keep running:

* PyDecoder   -- the reference's constructor and decode methods
                 (src/python_vali/src/PyDecoder.cpp:77-124, 310-346, 563-680) over two sources:
                 (a) raw video: the FFmpeg rawvideo options ({"f": "rawvideo", "video_size": "WxH",
                 "pixel_format": "nv12"|"yuv420p"|"p010le"|"yuv420p10le"}) or a ``.nv12`` / ``.yuv`` /
                 ``.p10`` suffix; (b) compressed input (mp4 / mkv / ...), when PyAV (`import av`, FFmpeg's
                 Python binding -- never vendored, absent from the build image) is importable on the box:
                 demux + decode on the CPU, planar -> semi-planar repack, upload into the caller's Surface
                 on the decoder's stream (north_star: "PyDecoder ... stubbed to CPU FFmpeg + hipMemcpy
                 upload").  Without PyAV compressed input raises RuntimeError.
* PyFrameConverter -- the reference's CPU (libswscale) converter API
                 (src/python_vali/src/PyFrameConverter.cpp:21-129) served by the HIP converter:
                 ndarray -> upload -> kernel -> download.  It is NOT a CPU code path.
* PyNvJpegEncoder -- the reference's JPEG encoder API on the CPU (download + Pillow) so that pipelines keep their
                 output side.
* PyNvEncoder  -- the reference's video encoder API (src/python_vali/src/PyNvEncoder.cpp:388-630) as "download + CPU
                 FFmpeg": the surface is downloaded on the encoder's stream and handed to a libavcodec encoder through
                 PyAV (libx264 / libx265 by default); raises RuntimeError only when PyAV is not importable.
* PacketData / SeekContext -- the plain data classes of VALI.cpp:216-279 the decoder calls take.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np

from .enums import ColorRange, ColorSpace, DecodeMode, FfmpegLogLevel, PixelFormat, TaskExecInfo
from .runtime import HipResMgr
from .surface import FORMATS, Surface
from .tasks import PySurfaceConverter
from .transfer import PyFrameUploader, PySurfaceDownloader

F = PixelFormat

# FFmpeg pix_fmt name -> (format in accelerated mode, format in CPU mode)
_PIX_FMTS = {
    "nv12": (F.NV12, F.NV12),
    "yuv420p": (F.NV12, F.YUV420),          # NVDEC always emits NV12; CPU decode emits planar
    "p010le": (F.P10, F.P10),
    "yuv420p10le": (F.P10, F.YUV420_10bit),
    "yuv444p": (F.YUV444, F.YUV444),
    "rgb24": (F.RGB, F.RGB),
}
_SUFFIX = {".nv12": "nv12", ".yuv": "yuv420p", ".yuv420": "yuv420p", ".p10": "p010le",
           ".yuv444": "yuv444p", ".rgb": "rgb24"}


def have_av() -> bool:
    """True when PyAV can be imported on this box (optional; nothing else in the package needs it)."""
    try:
        import av  # noqa: F401
        return True
    except Exception:
        return False


def SetFFMpegLogLevel(level) -> None:
    """reference: VALI.cpp:512-521 (`av_log_set_level(int(level))`).  Validates its argument like the pybind11 enum
    would (TypeError for anything that is not a FfmpegLogLevel) and forwards to PyAV's logging when PyAV is importable;
    without FFmpeg on the box there is no log to configure and the call is a no-op."""
    if not isinstance(level, FfmpegLogLevel):
        raise TypeError("SetFFMpegLogLevel(level: FfmpegLogLevel)")
    try:
        import av.logging as avlog
        avlog.set_level(int(level))
    except Exception:
        pass


class StreamParams:
    """reference: StreamParams of src/TC/inc/CodecsSupport.hpp, filled by GetStreamParams (TaskDecodeFrame.cpp:786-825);
    what PyDecoder.Probe returns, one per video stream."""

    __slots__ = ("width", "height", "fourcc", "codec_id", "color_space", "color_range", "num_frames", "start_time", "bit_rate",
                 "profile", "level", "fps", "avg_fps", "time_base", "start_time_sec", "duration_sec")

    def __init__(self):
        for name in self.__slots__:
            setattr(self, name, 0)
        self.color_space, self.color_range = ColorSpace.UNSPEC, ColorRange.UDEF
        self.fps = self.avg_fps = self.time_base = self.start_time_sec = self.duration_sec = 0.0

    def __repr__(self) -> str:
        return "StreamParams(" + ", ".join(f"{n}={getattr(self, n)!r}" for n in self.__slots__) + ")"


class PacketData:
    """Video frame metadata container (reference: VALI.cpp:250-279, MemoryInterfaces.hpp PacketData)."""

    __slots__ = ("key", "pts", "dts", "pos", "bsl", "duration")

    def __init__(self):
        self.key = self.pts = self.dts = self.pos = self.bsl = self.duration = 0

    def __repr__(self) -> str:
        return "".join(f"{name + ':':<10}{getattr(self, name)}\n" for name in self.__slots__)


class SeekContext:
    """SeekContext(seek_frame: int) / SeekContext(seek_ts: float) (reference: VALI.cpp:216-248): the type of the
    argument selects frame- or timestamp-based seeking, exactly like the two pybind11 constructors."""

    def __init__(self, seek_frame=None, seek_ts=None):
        if seek_ts is None and isinstance(seek_frame, float):
            seek_frame, seek_ts = None, seek_frame
        if (seek_frame is None) == (seek_ts is None):
            raise TypeError("SeekContext(seek_frame: int) or SeekContext(seek_ts: float)")
        self.seek_frame = -1 if seek_frame is None else int(seek_frame)
        self.seek_tssec = -1.0 if seek_ts is None else float(seek_ts)

    @property
    def IsByNumber(self) -> bool:
        return self.seek_frame >= 0

    @property
    def IsByTimestamp(self) -> bool:
        return self.seek_tssec >= 0.0


def _frame_to_flat(frame, av_fmt: str, w: int, h: int) -> np.ndarray:
    """One decoded PyAV frame -> the flat uint8 layout of `av_fmt` (yuv420p / yuv420p10le / nv12 / p010le).  Built from
    the PLANES of the planar form (every PyAV release exposes those), not from to_ndarray(format=...): several releases
    reject the semi-planar and 10-bit names there.  Same colourspace in and out: a byte shuffle, never a colour conversion."""
    ten = "10" in av_fmt
    planar = "yuv420p10le" if ten else "yuv420p"
    dt = np.uint16 if ten else np.uint8
    if hasattr(frame, "reformat") and hasattr(frame, "planes"):
        fr = frame.reformat(format=planar) if getattr(getattr(frame, "format", None), "name", planar) != planar else frame
        planes = []
        for i, pl in enumerate(fr.planes):
            pw, ph = (w, h) if i == 0 else (w // 2, h // 2)
            a = np.frombuffer(pl, np.uint8).reshape(ph, -1)[:, :pw * dt().itemsize]     # line_size >= row bytes
            planes.append(np.ascontiguousarray(a).view(dt).reshape(-1))
        y, u, v = planes
    else:                                       # minimal frame objects (tests): the planar array itself
        a = np.ascontiguousarray(frame.to_ndarray(format=planar)).view(dt).reshape(-1)
        y, u, v = a[:w * h], a[w * h:w * h + w * h // 4], a[w * h + w * h // 4:]
    if av_fmt in ("yuv420p", "yuv420p10le"):
        return np.concatenate([y, u, v]).view(np.uint8)
    uv = np.empty(u.size + v.size, dt)
    uv[0::2], uv[1::2] = u, v
    out = np.concatenate([y, uv])
    if ten:                                     # P010: the 10 bits sit in the high bits of each 16-bit word
        out = (out.astype(np.uint16) << 6).astype(np.uint16)
    return out.view(np.uint8)


# FF_PROFILE_* of the codecs this layer meets (libavcodec/defs.h): PyAV reports a profile by NAME, the reference by codecpar->profile
_FF_PROFILES = {"baseline": 66, "constrained baseline": 66 | (1 << 9), "main": 77, "extended": 88, "high": 100, "high 10": 110,
                "high 4:2:2": 122, "high 4:4:4 predictive": 244, "main 10": 2, "rext": 4, "profile 0": 0, "profile 1": 1,
                "profile 2": 2, "profile 3": 3, "simple": 0, "advanced simple": 15}


def _av_profile(cc) -> int:
    """codecpar->profile as an int (TaskDecodeFrame.cpp GetStreamParams): PyAV's own int when it has one, the FF_PROFILE value of
    its profile NAME when that is known (HEVC's "Main" is 1, not H.264's 77), otherwise 0 -- the same answer from the decoder and
    from Probe."""
    prof = getattr(cc, "profile", None)
    if isinstance(prof, int):
        return prof
    if not isinstance(prof, str):
        return 0
    name = prof.strip().lower()
    codec = str(getattr(getattr(cc, "codec", None), "name", "") or getattr(cc, "name", "")).lower()
    if codec in ("hevc", "h265") and name in ("main", "main 10", "main still picture", "rext"):
        return {"main": 1, "main 10": 2, "main still picture": 3, "rext": 4}[name]
    return _FF_PROFILES.get(name, 0)


def _av_base_rate(st):
    """the stream's r_frame_rate (PyAV: base_rate), else av_guess_frame_rate, else the average rate -- ONE order for the decoder's
    Framerate / IsVFR and for Probe().fps (TaskDecodeFrame.cpp:818, 908-922)"""
    return getattr(st, "base_rate", None) or getattr(st, "guessed_rate", None) or getattr(st, "average_rate", None)


class _AvSource:
    """Compressed input through PyAV: CPU demux + decode, one frame at a time, as flat uint8 arrays in the
    layout the reference's decoder emits -- planar YUV420[_10bit] in CPU mode, NV12 / P10 in accelerated mode
    (tests/test_PySurfaceUD.py:76-79,140-143; TaskDecodeFrame.cpp:575-650)."""

    _SPACE = {1: ColorSpace.BT_709, 5: ColorSpace.BT_601, 6: ColorSpace.BT_601}     # AVCOL_SPC_BT709 / BT470BG / SMPTE170M
    _RANGE = {1: ColorRange.MPEG, 2: ColorRange.JPEG}                                # AVCOL_RANGE_MPEG / JPEG

    def __init__(self, path: str, opts: dict, accelerated: bool):
        import av

        self._container = av.open(path, options={k: str(v) for k, v in opts.items()})
        self._stream = self._container.streams.video[0]
        cc = self._stream.codec_context
        self.width, self.height = int(cc.width), int(cc.height)
        pix = str(getattr(cc, "pix_fmt", None) or self._stream.format.name)
        self.high_bit_depth = "10" in pix or "12" in pix
        if accelerated:
            self.fmt, self._av_fmt = (F.P10, "p010le") if self.high_bit_depth else (F.NV12, "nv12")
        else:
            self.fmt, self._av_fmt = (F.YUV420_10bit, "yuv420p10le") if self.high_bit_depth else (F.YUV420, "yuv420p")
        rate = self._stream.average_rate or self._stream.guessed_rate
        self.framerate = float(rate) if rate else 25.0
        self.num_frames = int(self._stream.frames or 0)
        self.color_space = self._SPACE.get(int(getattr(cc, "colorspace", 2) or 2), ColorSpace.UNSPEC)
        self.color_range = self._RANGE.get(int(getattr(cc, "color_range", 0) or 0), ColorRange.UDEF)
        # r_frame_rate (TaskDecodeFrame.cpp:908-922) is PyAV's base_rate; guessed_rate (av_guess_frame_rate) may fall back to the
        # average or the codec rate and is only the stand-in when base_rate is missing
        base = _av_base_rate(self._stream)
        self.r_framerate = float(base) if base else self.framerate                 # != avg rate <=> VFR
        tb = getattr(self._stream, "time_base", None)
        self.time_base = float(tb) if tb else 0.0
        st = getattr(self._stream, "start_time", None)
        self.start_time = None if st is None else int(st)                          # AV_NOPTS_VALUE <=> None in PyAV
        dur = getattr(self._stream, "duration", None)
        self.duration = float(dur * tb) if (dur and tb) else 0.0
        self.bit_rate = int(getattr(cc, "bit_rate", 0) or getattr(self._container, "bit_rate", 0) or 0)
        self.gop_size = int(getattr(cc, "gop_size", 0) or 0)
        self.delay = int(getattr(cc, "delay", 0) or 0)
        self.profile, self.level = _av_profile(cc), int(getattr(cc, "level", 0) or 0)
        self.num_streams = len(getattr(self._container.streams, "video", [])) if not hasattr(self._container.streams, "__len__") \
            else len(self._container.streams)
        self.stream_index = int(getattr(self._stream, "index", 0) or 0)
        self.mode = DecodeMode.ALL_FRAMES
        self._frames = self._container.decode(self._stream)
        self.last_error = None
        self.last_frame = None      # the PyAV frame behind the most recent read() (pts / key frame for PacketData)
        if "12" in pix:
            import warnings
            warnings.warn("PyDecoder: 12-bit source is delivered as 10-bit (P10 / YUV420_10bit), like the reference's "
                          "decoder surfaces")

    @property
    def is_vfr(self) -> bool:
        return self.r_framerate != self.framerate                                  # TaskDecodeFrame.cpp:922

    def set_mode(self, mode: DecodeMode) -> None:
        """KEY_FRAMES: the codec skips everything but key frames (AVDISCARD_NONKEY, TaskDecodeFrame.cpp SetMode)."""
        want = "NONKEY" if DecodeMode(mode) == DecodeMode.KEY_FRAMES else "DEFAULT"
        cc = self._stream.codec_context
        err = None
        for value in self._skip_values(want):          # the SkipType enum of recent PyAV first, then the plain name
            try:
                cc.skip_frame = value
                err = None
                break
            except Exception as e:                      # noqa: BLE001 -- a setter that rejects this spelling
                err = e
        if err is not None:                             # never report a mode the codec is not in
            raise RuntimeError(f"PyDecoder.SetMode: this PyAV cannot set skip_frame={want}: {err}")
        self.mode = DecodeMode(mode)

    @staticmethod
    def _skip_values(name: str):
        try:
            import av
            skip_type = getattr(getattr(av.codec, "context", None), "SkipType", None)
            if skip_type is not None and hasattr(skip_type, name):
                yield getattr(skip_type, name)
        except Exception:                               # noqa: BLE001
            pass
        yield name

    def metadata(self) -> dict:
        """{"context": {...}, "video_stream": {...}} like GetMetaData (TaskDecodeFrame.cpp:846-870)"""
        out = {}
        for name, obj in (("context", self._container), ("video_stream", self._stream)):
            md = getattr(obj, "metadata", None)
            if md:
                out[name] = {str(k): str(v) for k, v in dict(md).items()}
        return out

    def seek_decode(self, ctx: "SeekContext") -> Tuple[Optional[np.ndarray], Optional[TaskExecInfo]]:
        """The reference's SeekDecode (TaskDecodeFrame.cpp:944-1029): seek BACKWARD to the key frame at or before the
        target, then decode and discard until the frame whose pts reaches the target -- that frame is the result
        (KEY_FRAMES mode: the key frame itself).  The target is in stream time-base units and includes the stream's
        start_time; seeking by frame number is refused on variable-frame-rate input (NOT_SUPPORTED)."""
        if self.is_vfr and ctx.IsByNumber:
            return None, TaskExecInfo.NOT_SUPPORTED
        if ctx.IsByNumber and not self.r_framerate > 0.0:      # no usable frame rate: a frame number has no timestamp
            return None, TaskExecInfo.NOT_SUPPORTED
        ts_sec = ctx.seek_frame / self.r_framerate if ctx.IsByNumber else ctx.seek_tssec
        tb = self.time_base or 1e-6
        target = int(round(ts_sec / tb))
        start = self.start_time if self.start_time is not None else 0
        target += start
        try:
            self._container.seek(target, stream=self._stream, backward=True, any_frame=False)
        except Exception as e:
            self.last_error = e
            return None, TaskExecInfo.FAIL
        self._frames = self._container.decode(self._stream)
        while True:
            data = self.read()
            if data is None:
                return None, None                        # end of stream / decode error: the caller reports which
            pts = getattr(self.last_frame, "pts", None)
            # frame pts are absolute stream timestamps, so the comparison is against the target WITH start_time.  (The
            # reference's loop reads `m_frame->pts + start_time < timestamp` with timestamp already offset: for streams
            # whose start_time is not 0 that stops start_time ticks early -- a known deviation, README "parity gaps".)
            # A frame without pts (AV_NOPTS_VALUE) keeps the loop going, as in the reference: NOPTS + start_time < timestamp.
            if self.mode == DecodeMode.KEY_FRAMES or (pts is not None and int(pts) >= target):
                return data, None

    def read(self) -> Optional[np.ndarray]:
        """next frame as a flat uint8 array, None at the end of the stream or on a decode error (last_error is set then)"""
        self.last_error = None
        try:
            frame = next(self._frames)
        except StopIteration:
            return None
        except Exception as e:       # av.error.* : invalid data, truncated file ...
            self.last_error = e
            return None
        self.last_frame = frame
        try:
            return _frame_to_flat(frame, self._av_fmt, self.width, self.height)
        except Exception as e:       # a pixel format this PyAV cannot reformat
            self.last_error = e
            return None

    def close(self):
        try:
            self._container.close()
        except Exception:
            pass


def _host_frame_size(fmt: PixelFormat, w: int, h: int) -> int:
    spec = FORMATS[fmt]
    return sum(pw * ph for pw, ph in spec.plane_geometry(w, h)) * spec.elem_size


class PyDecoder:
    """CPU-side stand-in for the FFmpeg/NVDEC decoder (same call surface): raw video always, compressed
    input when PyAV is importable.

    gpu_id >= 0: "accelerated" -- only DecodeSingleSurface[Async] works (PyDecoder.cpp:98-123);
    gpu_id <  0: CPU           -- only DecodeSingleFrame works (PyDecoder.cpp:77-96).
    """

    def __init__(self, input, opts: Optional[dict] = None, gpu_id: int = 0, stream=None):
        opts = dict(opts or {})
        self._gpu_id = int(gpu_id)
        self._mode = DecodeMode.ALL_FRAMES
        path = os.fspath(input) if isinstance(input, (str, os.PathLike)) else None
        if path is None:
            raise RuntimeError("PyDecoder: only file paths are supported by this build")
        suffix = os.path.splitext(path)[1].lower()
        pix = opts.get("pixel_format") or opts.get("pix_fmt") or _SUFFIX.get(suffix)
        self._av = None
        if opts.get("f", "rawvideo") != "rawvideo" or pix is None:
            if not have_av():
                raise RuntimeError(
                    "PyDecoder: compressed input needs PyAV (`import av`, FFmpeg's Python binding), which is not "
                    "importable here; the library itself has no demuxer / video decoder (the reference uses FFmpeg "
                    "+ NVDEC).  Raw video always works: opts={'f': 'rawvideo', 'video_size': 'WxH', "
                    "'pixel_format': 'nv12'} or a .nv12/.yuv/.p10 file with 'video_size'.")
            self._av = _AvSource(path, {k: v for k, v in opts.items() if k not in ("pixel_format", "pix_fmt")},
                                 self._gpu_id >= 0)
            self._w, self._h, self._fmt = self._av.width, self._av.height, self._av.fmt
            self._framerate, self._num_frames = self._av.r_framerate, self._av.num_frames
            self._file, self._pos = None, 0
            self._init_stream(stream)
            return
        if pix not in _PIX_FMTS:
            raise RuntimeError(f"PyDecoder: unsupported raw pixel_format {pix!r}")
        size = opts.get("video_size") or opts.get("s")
        if not size or "x" not in str(size):
            raise RuntimeError("PyDecoder: raw video needs opts['video_size'] = 'WxH'")
        self._w, self._h = (int(v) for v in str(size).lower().split("x"))
        self._file_fmt = pix
        self._fmt = _PIX_FMTS[pix][0 if self._gpu_id >= 0 else 1]
        self._framerate = float(opts.get("framerate", 25.0))
        self._file = open(path, "rb")
        self._file_frame = _host_frame_size(_PIX_FMTS[pix][1], self._w, self._h)
        self._num_frames = os.path.getsize(path) // self._file_frame
        self._pos = 0
        self._init_stream(stream)

    def _init_stream(self, stream):
        if self._gpu_id >= 0:
            self._stream = int(stream) if stream is not None else HipResMgr.Instance().GetStream(self._gpu_id)
            self._uploader = PyFrameUploader(self._gpu_id, self._stream)
        else:
            self._stream = 0

    # -- properties (PyDecoder.cpp:563-680) ------------------------------------------------
    Width = property(lambda self: self._w)
    Height = property(lambda self: self._h)
    Format = property(lambda self: self._fmt)
    Stream = property(lambda self: self._stream)
    NumFrames = property(lambda self: self._num_frames)
    Framerate = property(lambda self: self._framerate)
    AvgFramerate = property(lambda self: self._av.framerate if self._av else self._framerate)
    IsAccelerated = property(lambda self: self._gpu_id >= 0)
    IsVFR = property(lambda self: bool(self._av.is_vfr) if self._av else False)
    DisplayRotation = property(lambda self: 361.0)           # "no display matrix" value
    # the read-only ancillaries of PyDecoder.cpp:563-680, answered from PyAV's stream / codec context or the raw reader
    Level = property(lambda self: self._av.level if self._av else 0)
    Profile = property(lambda self: self._av.profile if self._av else None)
    Delay = property(lambda self: self._av.delay if self._av else 0)
    GopSize = property(lambda self: self._av.gop_size if self._av else 1)              # raw video: every frame is a key frame
    Bitrate = property(lambda self: self._av.bit_rate if self._av
                       else int(self._file_frame * 8 * self._framerate))
    NumStreams = property(lambda self: self._av.num_streams if self._av else 1)
    StreamIndex = property(lambda self: self._av.stream_index if self._av else 0)
    Timebase = property(lambda self: self._av.time_base if self._av else 1.0 / self._framerate)
    StartTime = property(lambda self: (self._av.start_time or 0) * self._av.time_base if self._av else 0.0)
    Duration = property(lambda self: self._av.duration if self._av else self._num_frames / self._framerate)
    MotionVectors = property(lambda self: [])                # needs the codec's side data: not exported by this stand-in
    Metadata = property(lambda self: self._av.metadata() if self._av else {})
    Mode = property(lambda self: self._mode)

    def SetMode(self, mode) -> None:
        """PyDecoder.cpp SetMode: KEY_FRAMES decodes key frames only.  Raw video: every frame is one."""
        if not isinstance(mode, DecodeMode):
            raise TypeError("SetMode(mode: DecodeMode)")
        self._mode = mode
        if self._av is not None:
            self._av.set_mode(mode)

    @staticmethod
    def Probe(input) -> list:  # noqa: A002 -- the reference's argument name
        """static Probe(input) -> list[StreamParams] (PyDecoder.cpp:684-700): the parameters of every video stream,
        without opening a codec.  Compressed input needs PyAV like the constructor; a raw file with a known suffix
        cannot be probed (it has no header) and raises RuntimeError like FFmpeg's 'Invalid data'."""
        path = os.fspath(input)
        if not have_av():
            raise RuntimeError("PyDecoder.Probe: needs PyAV (`import av`), which is not importable here")
        import av

        out = []
        with av.open(path) as container:
            for st in container.streams.video:
                cc, p = st.codec_context, StreamParams()
                p.width, p.height = int(cc.width), int(cc.height)
                p.codec_id = int(getattr(getattr(cc, "codec", None), "id", 0) or 0)
                tag = getattr(cc, "codec_tag", 0)
                p.fourcc = int.from_bytes(tag.encode("ascii", "replace")[:4].ljust(4, b"\0"), "little") if isinstance(tag, str) else int(tag or 0)
                p.color_space = _AvSource._SPACE.get(int(getattr(cc, "colorspace", 2) or 2), ColorSpace.UNSPEC)
                p.color_range = _AvSource._RANGE.get(int(getattr(cc, "color_range", 0) or 0), ColorRange.UDEF)
                p.num_frames, p.start_time = int(st.frames or 0), int(st.start_time or 0)
                p.bit_rate, p.level = int(getattr(cc, "bit_rate", 0) or 0), int(getattr(cc, "level", 0) or 0)
                p.profile = _av_profile(cc)
                rate, avg, tb = _av_base_rate(st), st.average_rate, st.time_base   # r_frame_rate, as GetStreamParams (TaskDecodeFrame.cpp:818)
                p.fps, p.avg_fps = float(rate or 0), float(avg or 0)
                p.time_base = float(tb) if tb else 0.0
                # the reference divides the stream-time-base values by AV_TIME_BASE (TaskDecodeFrame.cpp:821-822); kept as is
                p.start_time_sec, p.duration_sec = p.start_time / 1e6, float(st.duration or 0) / 1e6
                out.append(p)
        return out
    ColorSpace = property(lambda self: self._av.color_space if self._av else ColorSpace.UNSPEC)   # raw video carries no tags
    ColorRange = property(lambda self: self._av.color_range if self._av else ColorRange.UDEF)
    HostFrameSize = property(lambda self: _host_frame_size(self._fmt, self._w, self._h))

    def _next(self, seek_ctx) -> Tuple[Optional[np.ndarray], Optional[TaskExecInfo]]:
        """the next frame, or the frame a seek lands on; (None, info) when there is none (info None: end / decode error)"""
        if seek_ctx is None:
            return self._read(), None
        if self._av is not None:
            return self._av.seek_decode(seek_ctx)
        n = seek_ctx.seek_frame if seek_ctx.IsByNumber else int(round(seek_ctx.seek_tssec * self._framerate))
        self._pos = max(0, min(int(n), self._num_frames))
        return self._read(), None

    def _fill(self, pkt_data) -> None:
        """PacketData of the frame just read (the CPU stand-in knows the presentation order only)"""
        if pkt_data is None:
            return
        fr = self._av.last_frame if self._av is not None else None
        if fr is not None:
            pkt_data.pts = int(getattr(fr, "pts", 0) or 0)
            pkt_data.dts = int(getattr(fr, "dts", pkt_data.pts) or pkt_data.pts)
            pkt_data.key = int(bool(getattr(fr, "key_frame", 0)))
        else:
            pkt_data.pts = pkt_data.dts = self._pos - 1
            pkt_data.key, pkt_data.duration = 1, 1
            pkt_data.pos, pkt_data.bsl = (self._pos - 1) * self._file_frame, self._file_frame

    def _end(self) -> TaskExecInfo:
        failed = self._av is not None and self._av.last_error is not None
        return TaskExecInfo.FAIL if failed else TaskExecInfo.END_OF_STREAM

    def _read(self) -> Optional[np.ndarray]:
        if self._av is not None:
            return self._av.read()
        if self._pos >= self._num_frames:
            return None
        self._file.seek(self._pos * self._file_frame)
        raw = np.frombuffer(self._file.read(self._file_frame), np.uint8)
        self._pos += 1
        want, have = self._fmt, _PIX_FMTS[self._file_fmt][1]
        if want == have:
            return raw
        # accelerated mode on a planar 4:2:0 file: repack to the semi-planar layout NVDEC emits
        e = FORMATS[have].elem_size
        a = raw.view(np.uint8 if e == 1 else np.uint16)
        y = self._w * self._h
        c = y // 4
        uv = np.empty(2 * c, a.dtype)
        uv[0::2], uv[1::2] = a[y:y + c], a[y + c:y + 2 * c]
        return np.concatenate([a[:y], uv]).view(np.uint8)

    def DecodeSingleFrame(self, frame: np.ndarray, pkt_data=None, seek_ctx=None
                          ) -> Tuple[bool, TaskExecInfo]:
        if self.IsAccelerated:
            return False, TaskExecInfo.FAIL
        data, why = self._next(seek_ctx)
        if data is None:
            return False, (why or self._end())
        self._fill(pkt_data)
        if frame.nbytes != data.nbytes:
            frame.resize((data.nbytes // frame.itemsize,), refcheck=False)
        frame.view(np.uint8).reshape(-1)[:] = data
        return True, TaskExecInfo.SUCCESS

    def DecodeSingleSurface(self, surf: Surface, pkt_data=None, seek_ctx=None
                            ) -> Tuple[bool, TaskExecInfo]:
        ok, info = self.DecodeSingleSurfaceAsync(surf, pkt_data, seek_ctx)
        return ok, info          # the uploader already synchronises its stream

    def DecodeSingleSurfaceAsync(self, surf: Surface, pkt_data=None, seek_ctx=None
                                 ) -> Tuple[bool, TaskExecInfo]:
        if not self.IsAccelerated:
            return False, TaskExecInfo.FAIL
        if surf is None or surf.IsEmpty:
            return False, TaskExecInfo.INVALID_INPUT
        if (surf.Width, surf.Height) != (self._w, self._h) or surf.Format != self._fmt:
            return False, TaskExecInfo.INVALID_INPUT
        data, why = self._next(seek_ctx)
        if data is None:
            return False, (why or self._end())
        self._fill(pkt_data)
        return self._uploader.Run(data, surf)

    def __del__(self):
        f = getattr(self, "_file", None)
        if f:
            f.close()
        a = getattr(self, "_av", None)
        if a is not None:
            a.close()


class PyFrameConverter:
    """ndarray -> ndarray pixel-format conversion with the reference's CPU-converter signature
    (PyFrameConverter.cpp:21-129), executed by the HIP converter on `gpu_id` (default 0)."""

    def __init__(self, width: int, height: int, src_format: PixelFormat, dst_format: PixelFormat,
                 gpu_id: int = 0):
        self._w, self._h = int(width), int(height)
        self._src_fmt, self._dst_fmt = PixelFormat(src_format), PixelFormat(dst_format)
        self._cvt = PySurfaceConverter(gpu_id)
        if (self._src_fmt, self._dst_fmt) not in PySurfaceConverter.Conversions():
            raise RuntimeError(f"Unsupported conversion {self._src_fmt.name} -> {self._dst_fmt.name}")
        self._src = Surface.Make(self._src_fmt, self._w, self._h, gpu_id)
        self._dst = Surface.Make(self._dst_fmt, self._w, self._h, gpu_id)
        self._up = PyFrameUploader(gpu_id, self._cvt.Stream)
        self._down = PySurfaceDownloader(gpu_id, self._cvt.Stream)

    @property
    def Format(self) -> PixelFormat:
        return self._dst_fmt

    def Run(self, src: np.ndarray, dst: np.ndarray, cc_ctx=None) -> Tuple[bool, TaskExecInfo]:
        if src.nbytes != self._src.HostSize:                     # PyFrameConverter.cpp:36-44
            return False, TaskExecInfo.INVALID_INPUT
        if dst.nbytes != self._dst.HostSize:
            dst.resize((self._dst.HostSize // dst.itemsize,), refcheck=False)
        ok, info = self._up.Run(src, self._src)
        if not ok:
            return ok, info
        ok, info = self._cvt.RunAsync(self._src, self._dst, cc_ctx)
        if not ok:
            return ok, info
        return self._down.Run(self._dst, dst)


class PyNvEncoder:
    """Video encode with the reference's call surface (src/python_vali/src/PyNvEncoder.cpp:388-630), as the north_star
    asks: "stubbed to CPU FFmpeg".  The surface is downloaded on the encoder's stream and compressed by libavcodec through
    PyAV; RuntimeError when PyAV is not importable (the library itself has no encoder).

    settings (strings, like the reference's): 's': 'WxH' (required), 'codec': 'h264' | 'hevc' (default h264), 'fps',
    'bitrate' (e.g. '5M'), 'gop', 'preset'; anything else is passed to the codec as a private option.
    EncodeSingleSurface(surface, packet[, sei][, sync][, append]) -> True when `packet` received bytes (encoders buffer
    frames: False is not an error); Flush(packets) drains."""

    _CODECS = {"h264": ("libx264", "h264"), "hevc": ("libx265", "hevc"), "h265": ("libx265", "hevc")}

    def __init__(self, settings, gpu_id: int, *args, format: PixelFormat = None, verbose: bool = False, stream=None):  # noqa: A002
        rest = list(args)
        if rest and not isinstance(rest[0], PixelFormat) and isinstance(rest[0], int) and stream is None:
            stream = rest.pop(0)                   # (settings, gpu_id, stream, format, verbose)
        if rest and format is None:
            format = rest.pop(0)                   # noqa: A001
        if rest:
            verbose = bool(rest.pop(0))
        self._fmt = PixelFormat(format) if format is not None else F.NV12
        if self._fmt not in (F.NV12, F.YUV420):
            raise RuntimeError(f"PyNvEncoder: unsupported input format {self._fmt.name} (NV12 or YUV420)")
        if not have_av():
            raise RuntimeError("PyNvEncoder: video encode needs PyAV (`import av`, FFmpeg's Python binding), which is not "
                               "importable here; encode ASICs are outside the surface-processing path")
        import av
        from fractions import Fraction

        st = {str(k): str(v) for k, v in dict(settings).items()}
        size = st.pop("s", None) or st.pop("video_size", None)
        if not size or "x" not in size.lower():
            raise RuntimeError("PyNvEncoder: settings['s'] = 'WxH' is required")
        self._w, self._h = (int(v) for v in size.lower().split("x"))
        names = self._CODECS.get(st.pop("codec", "h264").lower())
        if names is None:
            raise RuntimeError("PyNvEncoder: codec must be h264 or hevc")
        self._names, self._fps, self._st, self._av = names, int(float(st.pop("fps", "30"))), st, av
        self._ctx = self._open()
        self._n, self._drained = 0, False
        self._gpu_id = int(gpu_id)
        self._stream = int(stream) if stream is not None else HipResMgr.Instance().GetStream(self._gpu_id)
        self._down = PySurfaceDownloader(self._gpu_id, self._stream)
        self._verbose = verbose

    def _open(self):
        """A fresh libavcodec encoder context from the stored settings.  Configured for ZERO DELAY (no B frames, no
        look-ahead, no frame threads; `tune=zerolatency` for libx264 / libx265 unless the caller chose a tune): NVENC's
        default presets hand every frame's packet back at once, and the reference's `sync` mode relies on it
        (PyNvEncoder.cpp:496-602).  An encoder that has been drained (Flush) is at EOF for good -- avcodec_send_frame
        returns AVERROR_EOF -- so the next frame after a Flush gets a new context from here."""
        from fractions import Fraction

        av, st = self._av, dict(self._st)
        ctx = None
        for name in self._names:
            try:
                ctx = av.CodecContext.create(name, "w")
                used = name
                break
            except Exception:           # this FFmpeg build lacks the encoder: try the next name
                continue
        if ctx is None:
            raise RuntimeError(f"PyNvEncoder: no {self._names[-1]} encoder in this FFmpeg build")
        fps = self._fps
        ctx.width, ctx.height = self._w, self._h
        ctx.pix_fmt = "yuv420p"
        ctx.time_base = Fraction(1, max(fps, 1))
        for attr, val in (("framerate", Fraction(max(fps, 1), 1)), ("max_b_frames", 0), ("thread_count", 1)):
            try:
                setattr(ctx, attr, val)
            except Exception:
                pass
        if "bitrate" in st:
            b = st.pop("bitrate").upper()
            ctx.bit_rate = int(float(b.rstrip("KM")) * (1000 if b.endswith("K") else 1000000 if b.endswith("M") else 1))
        if "gop" in st:
            ctx.gop_size = int(st.pop("gop"))
        if used.startswith("libx26"):
            st.setdefault("tune", "zerolatency")
        ctx.options = st
        return ctx

    Width = property(lambda self: self._w)
    Height = property(lambda self: self._h)
    Format = property(lambda self: self._fmt)
    FrameSizeInBytes = property(lambda self: _host_frame_size(self._fmt, self._w, self._h))
    Capabilities = property(lambda self: {})       # NV_ENC_CAPS of an NVENC session: none on this backend

    @staticmethod
    def _emit(packets, out: np.ndarray, append: bool) -> bool:
        data = b"".join(bytes(p) for p in packets)
        if not data and not append:
            out.resize((0,), refcheck=False)
            return False
        new = np.frombuffer(data, np.uint8)
        keep = out.copy() if append else out[:0].copy()
        out.resize((keep.size + new.size,), refcheck=False)
        out[:keep.size], out[keep.size:] = keep, new
        return bool(data)

    def EncodeSingleSurface(self, surface: Surface, packet: np.ndarray, *args, sei=None, sync=False, append=False) -> bool:
        """EncodeSingleSurface(surface, packet[, sei][, sync][, append]): the reference's five overloads
        (PyNvEncoder.cpp:496-602); `sei` is accepted and ignored (no SEI insertion through libavcodec here)."""
        rest = list(args)
        if rest and isinstance(rest[0], np.ndarray):
            sei = rest.pop(0)
        if rest:
            sync = bool(rest.pop(0))
        if rest:
            append = bool(rest.pop(0))
        if surface is None or surface.IsEmpty or surface.Format != self._fmt or \
                (surface.Width, surface.Height) != (self._w, self._h):
            return False
        host = np.zeros(surface.HostSize, np.uint8)
        if not self._down.Run(surface, host)[0]:
            return False
        w, h = self._w, self._h
        if self._fmt == F.NV12:                    # -> planar 4:2:0, the one layout every libavcodec encoder takes
            y, uv = host[:w * h], host[w * h:]
            host = np.concatenate([y, uv[0::2], uv[1::2]])
        frame = self._av.VideoFrame.from_ndarray(host.reshape(h * 3 // 2, w), format="yuv420p")
        frame.pts = self._n
        self._n += 1
        if self._drained:                          # a drained libavcodec encoder is at EOF: start a new one
            self._ctx, self._drained = self._open(), False
        # zero-delay configuration (_open): this frame's packet comes back from this call, `sync` or not; an encoder that
        # still holds frames back (a codec that ignores the settings) simply returns False -- never drained per frame,
        # which would put it in EOF state for the next one
        return self._emit(self._ctx.encode(frame), packet, append)

    def Flush(self, packets: np.ndarray) -> bool:
        if self._drained:
            return self._emit([], packets, False)
        self._drained = True
        return self._emit(self._ctx.encode(None), packets, False)

    def FlushSinglePacket(self, packets: np.ndarray) -> bool:
        return self.Flush(packets)

    def Reconfigure(self, settings, force_idr: bool = False, reset_encoder: bool = False, verbose: bool = False) -> bool:
        return False                               # libavcodec contexts are not reconfigurable in flight


class NvJpegEncodeContext:
    """reference: NvJpegEncodeContext (src/TC/inc/Tasks.hpp:251-267, src/TC/src/TaskNvJpegEncode.cpp:93-124):
    compression coefficient + the pixel format of the surfaces to encode."""

    # format -> PIL JPEG `subsampling` (TaskNvJpegEncode.cpp:101-124: RGB inputs encode 4:4:4, YUV inputs keep theirs)
    _SUBSAMPLING = {F.RGB: 0, F.BGR: 0, F.RGB_PLANAR: 0, F.YUV444: 0, F.YUV422: 1, F.YUV420: 2}

    def __init__(self, compression: int = 100, pixel_format: PixelFormat = F.RGB):
        fmt = PixelFormat(pixel_format)
        if fmt not in self._SUBSAMPLING:
            raise ValueError("unsupported pixel format")            # std::invalid_argument, :123
        self._compression, self._format = int(compression), fmt

    def Compression(self) -> int:
        return self._compression

    def Format(self) -> PixelFormat:
        return self._format


class PyNvJpegEncoder:
    """JPEG encoder with the reference's call surface (src/python_vali/src/PyNvJpegEncoder.cpp:21-160), served on the
    CPU: the reference uses the nvJPEG ASIC / CUDA library, this backend downloads the surface on the encoder's stream
    and compresses with Pillow (libjpeg) when it is importable -- the output side of pipelines written against
    python_vali keeps working (north_star: decode / encode classes are CPU stubs around the surface path).
    `Run(context, surfaces)` -> (list of uint8 arrays, TaskExecInfo): all surfaces or none (:36-75)."""

    def __init__(self, gpu_id: int):
        try:
            from PIL import Image  # noqa: F401
        except Exception as exc:  # pragma: no cover - depends on the environment
            raise RuntimeError("PyNvJpegEncoder: no JPEG ASIC on this backend and Pillow (PIL) is not importable for the "
                               f"CPU fallback ({exc})") from exc
        self._gpu_id = int(gpu_id)
        self._stream = HipResMgr.Instance().GetStream(self._gpu_id)
        self._down = PySurfaceDownloader(self._gpu_id, self._stream)

    def Context(self, compression: int, pixel_format: PixelFormat) -> NvJpegEncodeContext:
        return NvJpegEncodeContext(compression, pixel_format)

    def _image(self, fmt: PixelFormat, w: int, h: int, host: np.ndarray):
        from PIL import Image

        if fmt == F.RGB:
            return Image.frombuffer("RGB", (w, h), host.tobytes(), "raw", "RGB", 0, 1)
        if fmt == F.BGR:
            return Image.fromarray(np.ascontiguousarray(host.reshape(h, w, 3)[..., ::-1]), "RGB")
        if fmt == F.RGB_PLANAR:
            return Image.fromarray(np.ascontiguousarray(host.reshape(3, h, w).transpose(1, 2, 0)), "RGB")
        # planar YUV: JPEG stores YCbCr as it is (no colour conversion, like NVJPEG_INPUT_YUV); chroma planes are
        # replicated to full size and libjpeg subsamples them again with the context's factors
        y = host[: w * h].reshape(h, w)
        cw, ch = (w, h) if fmt == F.YUV444 else (w // 2, h) if fmt == F.YUV422 else (w // 2, h // 2)
        u = host[w * h: w * h + cw * ch].reshape(ch, cw)
        v = host[w * h + cw * ch: w * h + 2 * cw * ch].reshape(ch, cw)
        if (cw, ch) != (w, h):
            u = np.repeat(np.repeat(u, h // ch, 0), w // cw, 1)[:h, :w]
            v = np.repeat(np.repeat(v, h // ch, 0), w // cw, 1)[:h, :w]
        return Image.fromarray(np.ascontiguousarray(np.stack([y, u, v], -1)), "YCbCr")

    def Run(self, context: NvJpegEncodeContext, surfaces) -> Tuple[list, TaskExecInfo]:
        import io

        buffers = []
        for surf in surfaces:
            if surf is None or surf.IsEmpty or surf.Format != context.Format():      # :41-45: all or nothing
                return [], TaskExecInfo.FAIL
            host = np.zeros(surf.HostSize, np.uint8)
            ok, _ = self._down.Run(surf, host)
            if not ok:
                return [], TaskExecInfo.FAIL
            out = io.BytesIO()
            q = max(1, min(100, context.Compression()))
            self._image(surf.Format, surf.Width, surf.Height, host).save(
                out, format="JPEG", quality=q, subsampling=NvJpegEncodeContext._SUBSAMPLING[surf.Format])
            buffers.append(np.frombuffer(out.getvalue(), np.uint8).copy())
        return buffers, TaskExecInfo.SUCCESS
