"""Enumerations of the python_vali API, same names and values as the reference.

reference: src/TC/inc/MemoryInterfaces.hpp:29-58 (Pixel_Format, ColorSpace, ColorRange),
src/TC/TC_CORE/inc/TC_CORE.hpp:38-52 (TaskExecInfo), src/python_vali/src/VALI.cpp:130-205
(pybind11 enums with export_values()).
"""
from __future__ import annotations

import enum


class PixelFormat(enum.IntEnum):
    UNDEFINED = 0
    Y = 1
    RGB = 2
    NV12 = 3
    YUV420 = 4
    RGB_PLANAR = 5
    BGR = 6
    YUV444 = 7
    RGB_32F = 8
    RGB_32F_PLANAR = 9
    YUV422 = 10
    P10 = 11
    P12 = 12
    YUV444_10bit = 13
    YUV420_10bit = 14
    GRAY12 = 15

    def __str__(self) -> str:  # pybind11 prints "PixelFormat.NV12"
        return f"PixelFormat.{self.name}"


class TaskExecInfo(enum.IntEnum):
    SUCCESS = 0
    FAIL = 1
    END_OF_STREAM = 2
    MORE_DATA_NEEDED = 3
    BIT_DEPTH_NOT_SUPPORTED = 4
    INVALID_INPUT = 5
    UNSUPPORTED_FMT_CONV_PARAMS = 6
    NOT_SUPPORTED = 7
    RES_CHANGE = 8
    SRC_DST_SIZE_MISMATCH = 9
    SRC_DST_FMT_MISMATCH = 10

    def __str__(self) -> str:
        return f"TaskExecInfo.{self.name}"


class TaskExecStatus(enum.IntEnum):
    TASK_EXEC_SUCCESS = 0
    TASK_EXEC_FAIL = 1


class ColorSpace(enum.IntEnum):
    BT_601 = 0
    BT_709 = 1
    UNSPEC = 2


class ColorRange(enum.IntEnum):
    MPEG = 0
    JPEG = 1
    UDEF = 2


class Interpolation(enum.IntEnum):
    """Resize filter (new: the reference hard-codes NPP Lanczos, TaskResizeSurface.cpp:67).
    Values = include/vali_hip.h `vali_interpolation` (NppiInterpolationMode numbering)."""
    LINEAR = 1
    CUBIC = 4
    LANCZOS = 16


class DecodeMode(enum.IntEnum):
    """reference: src/TC/inc/CodecsSupport.hpp:148, binding VALI.cpp:184-187."""
    KEY_FRAMES = 0
    ALL_FRAMES = 1


class FfmpegLogLevel(enum.IntEnum):
    """reference: VALI.cpp:48-56, 206-214 (the AV_LOG_* values of libavutil/log.h)."""
    PANIC = 0
    FATAL = 8
    ERROR = 16
    WARNING = 24
    INFO = 32
    VERBOSE = 40
    DEBUG = 48


class DLDeviceType(enum.IntEnum):
    """DLPack device types.  The reference exports kDLCUDA only
    (src/TC/src/SurfacePlane.cpp:255); on ROCm the exchange type is kDLROCM."""

    kDLCPU = 1
    kDLCUDA = 2
    kDLCUDAHost = 3
    kDLROCM = 10
    kDLROCMHost = 11
    kDLCUDAManaged = 13


class ColorspaceConversionContext:
    """reference: src/TC/inc/MemoryInterfaces.hpp:60-68, binding VALI.cpp:326-348."""

    __slots__ = ("color_space", "color_range")

    def __init__(self, color_space: ColorSpace = ColorSpace.UNSPEC,
                 color_range: ColorRange = ColorRange.UDEF):
        self.color_space = ColorSpace(color_space)
        self.color_range = ColorRange(color_range)

    def __repr__(self) -> str:
        return f"ColorspaceConversionContext({self.color_space.name}, {self.color_range.name})"


class TaskExecDetails:
    """reference: src/TC/TC_CORE/inc/TC_CORE.hpp:54-67."""

    __slots__ = ("status", "info", "message")

    def __init__(self, status=TaskExecStatus.TASK_EXEC_SUCCESS, info=TaskExecInfo.SUCCESS,
                 message: str = ""):
        self.status = TaskExecStatus(status)
        self.info = TaskExecInfo(info)
        self.message = message

    @classmethod
    def ok(cls) -> "TaskExecDetails":
        return cls()

    @classmethod
    def failed(cls, info: TaskExecInfo, message: str = "") -> "TaskExecDetails":
        return cls(TaskExecStatus.TASK_EXEC_FAIL, info, message)

    @property
    def success(self) -> bool:
        return self.status == TaskExecStatus.TASK_EXEC_SUCCESS

    def __repr__(self) -> str:
        return f"TaskExecDetails({self.status.name}, {self.info.name}, {self.message!r})"


def export_values(namespace: dict) -> None:
    """pybind11's export_values(): enum members become module attributes."""
    for en in (PixelFormat, TaskExecInfo, ColorSpace, ColorRange, DLDeviceType, DecodeMode, FfmpegLogLevel):
        for member in en:
            namespace[member.name] = member
