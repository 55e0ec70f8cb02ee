"""CudaBuffer: a flat device buffer (reference: src/TC/inc/MemoryInterfaces.hpp:119-151,
src/TC/src/MemoryInterfaces.cpp CudaBuffer::*, Python binding src/python_vali/src/VALI.cpp:349-441).

Same names and behaviour: Make(elem_size, num_elems, gpu_id); RawMemSize / NumElems / ElemSize /
GpuMem; Clone() copies on the GPU's default stream and waits; CopyFrom(other, stream|gpu_id)
raises RuntimeError when sizes differ, copies device-to-device and synchronises the stream
(VALI.cpp:29-47)."""
from __future__ import annotations

from ._native import shim
from .runtime import HipResMgr
from .surface import _DeviceMem


class CudaBuffer:
    def __init__(self, *args, **kwargs):
        raise TypeError("CudaBuffer: use CudaBuffer.Make(elem_size, num_elems, gpu_id)")

    @classmethod
    def _new(cls, elem_size: int, num_elems: int, gpu_id: int) -> "CudaBuffer":
        self = object.__new__(cls)
        self._elem_size, self._num_elems, self._gpu_id = int(elem_size), int(num_elems), int(gpu_id)
        if self._elem_size <= 0 or self._num_elems <= 0:
            raise RuntimeError("CudaBuffer: elem_size and num_elems must be positive")
        self._mem = _DeviceMem(shim.mem_alloc(self._gpu_id, self._elem_size * self._num_elems), self._gpu_id)
        return self

    @staticmethod
    def Make(elem_size: int, num_elems: int, gpu_id: int) -> "CudaBuffer":
        return CudaBuffer._new(elem_size, num_elems, gpu_id)

    @property
    def RawMemSize(self) -> int:
        return self._elem_size * self._num_elems

    @property
    def NumElems(self) -> int:
        return self._num_elems

    @property
    def ElemSize(self) -> int:
        return self._elem_size

    @property
    def GpuMem(self) -> int:
        return self._mem.ptr

    def _copy_from(self, other: "CudaBuffer", stream: int) -> None:
        if not isinstance(other, CudaBuffer):
            raise TypeError("CopyFrom: other must be a CudaBuffer")
        if other.RawMemSize != self.RawMemSize:                  # VALI.cpp:31-33
            raise RuntimeError("Can't copy: buffers have different size.")
        n = self.RawMemSize
        shim.memcpy2d_async(self._gpu_id, self.GpuMem, n, other.GpuMem, n, n, 1, 2, stream)
        shim.stream_sync(self._gpu_id, stream)

    def CopyFrom(self, other: "CudaBuffer", stream: int = None, gpu_id: int = None) -> None:
        """CopyFrom(other, stream) / CopyFrom(other, gpu_id=...).  As with the reference's two
        pybind11 overloads, a positional integer is taken as the stream handle."""
        if stream is None:
            stream = HipResMgr.Instance().GetStream(self._gpu_id if gpu_id is None else int(gpu_id))
        self._copy_from(other, int(stream))

    def Clone(self) -> "CudaBuffer":
        c = CudaBuffer._new(self._elem_size, self._num_elems, self._gpu_id)
        c._copy_from(self, HipResMgr.Instance().GetStream(self._gpu_id))
        return c

    def __repr__(self) -> str:
        return f"CudaBuffer(elem_size={self._elem_size}, num_elems={self._num_elems}, gpu_id={self._gpu_id})"
