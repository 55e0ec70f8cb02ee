"""Host <-> device frame transfer: PyFrameUploader / PySurfaceDownloader.

reference: src/TC/src/TaskCudaUploadFrame.cpp:28-82, src/TC/src/TaskCudaDownloadSurface.cpp:28-82,
bindings src/python_vali/src/PyFrameUploader.cpp:33-99, PySurfaceDownloader.cpp:35-100.
One 2-D copy per plane between a tightly packed host buffer and the pitched planes,
then a stream sync (both classes are blocking in the reference).
"""
from __future__ import annotations

from typing import Tuple

from ._native import shim
from .enums import TaskExecInfo
from .runtime import HipResMgr
from .surface import Surface


class _Transfer:
    def __init__(self, gpu_id: int, stream=None):
        self._gpu_id = int(gpu_id)
        self._stream = int(stream) if stream is not None else HipResMgr.Instance().GetStream(self._gpu_id)

    @property
    def Stream(self) -> int:
        return self._stream

    def _copy(self, surf: Surface, host, to_device: bool) -> Tuple[bool, TaskExecInfo]:
        if surf is None or surf.IsEmpty:
            return False, TaskExecInfo.INVALID_INPUT
        try:
            ptr, nbytes, contiguous = shim.buffer_info(host, not to_device)
        except Exception:
            return False, TaskExecInfo.INVALID_INPUT
        if not contiguous:
            return False, TaskExecInfo.INVALID_INPUT
        if nbytes != surf.HostSize:
            return False, TaskExecInfo.SRC_DST_SIZE_MISMATCH
        try:
            off = 0
            for p in surf._planes:
                row = p.Width * p.ElemSize
                if to_device:
                    shim.memcpy2d_async(self._gpu_id, p.GpuMem, p.Pitch, ptr + off, row, row,
                                        p.Height, 0, self._stream)
                else:
                    shim.memcpy2d_async(self._gpu_id, ptr + off, row, p.GpuMem, p.Pitch, row,
                                        p.Height, 1, self._stream)
                off += row * p.Height
            shim.stream_sync(self._gpu_id, self._stream)
        except RuntimeError:
            return False, TaskExecInfo.FAIL
        return True, TaskExecInfo.SUCCESS


class PyFrameUploader(_Transfer):
    """numpy ndarray -> Surface (blocking)."""

    def Run(self, src, dst: Surface) -> Tuple[bool, TaskExecInfo]:
        return self._copy(dst, src, True)


class PySurfaceDownloader(_Transfer):
    """Surface -> numpy ndarray (blocking)."""

    def Run(self, src: Surface, dst) -> Tuple[bool, TaskExecInfo]:
        return self._copy(src, dst, False)
