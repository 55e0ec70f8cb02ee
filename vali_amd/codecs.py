"""Callers on either side of the surface path: decoder / encoder / CPU-converter classes.

The reference implements these with FFmpeg + NVDEC/NVENC/nvJPEG (SURVEY.md section 2 rows
12-15): decode/encode ASICs and a demuxer are outside this build's scope and no FFmpeg exists
offline.  What IS provided so pipelines written against python_vali keep running:

* PyDecoder   -- raw-video reader with the reference's constructor and decode methods
                 (src/python_vali/src/PyDecoder.cpp:77-124, 310-346, 563-680): it accepts the
                 FFmpeg rawvideo options ({"f": "rawvideo", "video_size": "WxH",
                 "pixel_format": "nv12"|"yuv420p"|"p010le"|"yuv420p10le"}) or infers them from a
                 ``.nv12`` / ``.yuv`` / ``.p10`` suffix, and uploads frames into the caller's
                 Surface on its stream.  Compressed input raises RuntimeError.
* PyFrameConverter -- the reference's CPU (libswscale) converter API
                 (src/python_vali/src/PyFrameConverter.cpp:21-129) served by the HIP converter:
                 ndarray -> upload -> kernel -> download.  It is NOT a CPU code path.
* PyNvEncoder / PyNvJpegEncoder -- raise: no encoder in scope.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np

from .enums import ColorRange, ColorSpace, PixelFormat, TaskExecInfo
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


def _host_frame_size(fmt: PixelFormat, w: int, h: int) -> int:
    spec = FORMATS[fmt]
    return sum(pw * ph for pw, ph in spec.plane_geometry(w, h)) * spec.elem_size


class PyDecoder:
    """Raw-video stand-in for the FFmpeg/NVDEC decoder (same call surface).

    gpu_id >= 0: "accelerated" -- only DecodeSingleSurface[Async] works (PyDecoder.cpp:98-123);
    gpu_id <  0: CPU           -- only DecodeSingleFrame works (PyDecoder.cpp:77-96).
    """

    def __init__(self, input, opts: Optional[dict] = None, gpu_id: int = 0, stream=None):
        opts = dict(opts or {})
        self._gpu_id = int(gpu_id)
        path = os.fspath(input) if isinstance(input, (str, os.PathLike)) else None
        if path is None:
            raise RuntimeError("PyDecoder: only file paths are supported by this build")
        suffix = os.path.splitext(path)[1].lower()
        pix = opts.get("pixel_format") or opts.get("pix_fmt") or _SUFFIX.get(suffix)
        if opts.get("f", "rawvideo") != "rawvideo" or pix is None:
            raise RuntimeError(
                "PyDecoder: this MI355X build has no demuxer / video decoder (the reference uses "
                "FFmpeg + NVDEC).  Raw video is supported: opts={'f': 'rawvideo', 'video_size': "
                "'WxH', 'pixel_format': 'nv12'} or a .nv12/.yuv/.p10 file with 'video_size'.")
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
    AvgFramerate = property(lambda self: self._framerate)
    IsAccelerated = property(lambda self: self._gpu_id >= 0)
    IsVFR = property(lambda self: False)
    DisplayRotation = property(lambda self: 361.0)           # "no display matrix" value
    ColorSpace = property(lambda self: ColorSpace.UNSPEC)     # raw video carries no tags
    ColorRange = property(lambda self: ColorRange.UDEF)
    HostFrameSize = property(lambda self: _host_frame_size(self._fmt, self._w, self._h))

    def _read(self) -> Optional[np.ndarray]:
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
        data = self._read()
        if data is None:
            return False, TaskExecInfo.END_OF_STREAM
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
        data = self._read()
        if data is None:
            return False, TaskExecInfo.END_OF_STREAM
        return self._uploader.Run(data, surf)

    def __del__(self):
        f = getattr(self, "_file", None)
        if f:
            f.close()


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


class _NoEncoder:
    def __init__(self, *args, **kwargs):
        raise RuntimeError(f"{type(self).__name__}: video / JPEG encode ASICs are not supported on "
                           "this backend (out of scope of the surface-processing path)")


class PyNvEncoder(_NoEncoder):
    """reference: src/python_vali/src/PyNvEncoder.cpp (NVENC)."""


class PyNvJpegEncoder(_NoEncoder):
    """reference: src/python_vali/src/PyNvJpegEncoder.cpp (nvJPEG)."""
