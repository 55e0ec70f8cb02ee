"""Surface / SurfacePlane: pitched device planes, DLPack + array-interface exchange.

Host-side mirror of the reference's L2 layer
(reference: src/TC/inc/SurfacePlane.hpp:52-285, src/TC/src/SurfacePlane.cpp,
src/TC/src/Surfaces.cpp, src/TC/src/MemoryInterfaces.cpp:330-480,
binding src/python_vali/src/PySurface.cpp:111-553).  The 14 C++ format classes
collapse into one table of plane geometries (FORMATS).

Device memory comes from libvali_hip.so (hipMalloc, 256-byte pitch policy); DLPack
tensors are exported as kDLROCM so torch.from_dlpack() works on PyTorch-ROCm.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

from ._native import shim
from .enums import DLDeviceType, PixelFormat

kDLUInt = 1
kDLFloat = 2


class _DeviceMem:
    """Owner of one hipMalloc allocation (the shared_ptr<void> of SurfacePlane.cpp:204-212)."""

    __slots__ = ("ptr", "device", "__weakref__")

    def __init__(self, ptr: int, device: int):
        self.ptr = ptr
        self.device = device

    def __del__(self):
        if getattr(self, "ptr", 0):
            try:
                shim.mem_free(self.device, self.ptr)
            except Exception:
                pass
            self.ptr = 0


class SurfacePlane:
    """One 2-D pitched chunk of device memory (a plane / channel group).

    Width is in ELEMENTS (so 3*W for a packed RGB plane), like the reference.
    """

    __slots__ = ("_w", "_h", "_pitch", "_elem", "_code", "_typestr", "_ptr", "_device",
                 "_owner", "_own", "_from_dlpack")

    def __init__(self, width=0, height=0, pitch=0, elem_size=0, type_code=kDLUInt,
                 typestr="<u1", ptr=0, device=0, owner=None, own=False, from_dlpack=False):
        self._w, self._h, self._pitch, self._elem = int(width), int(height), int(pitch), int(elem_size)
        self._code, self._typestr = type_code, typestr
        self._ptr, self._device = int(ptr), int(device)
        self._owner, self._own, self._from_dlpack = owner, own, from_dlpack

    # -- construction -----------------------------------------------------------
    @classmethod
    def allocate(cls, width: int, height: int, elem_size: int, type_code: int, typestr: str,
                 device: int) -> "SurfacePlane":
        """cuMemAllocPitch analogue (SurfacePlane.cpp:186-213)."""
        ptr, pitch = shim.mem_alloc_pitch(device, width * elem_size, height)
        mem = _DeviceMem(ptr, device)
        return cls(width, height, pitch, elem_size, type_code, typestr, ptr, device, mem, True)

    def view(self) -> "SurfacePlane":
        """Non-owning copy (copy-ctor semantics, SurfacePlane.cpp:52-55); keeps the
        allocation alive."""
        return SurfacePlane(self._w, self._h, self._pitch, self._elem, self._code, self._typestr,
                            self._ptr, self._device, self._owner, False, self._from_dlpack)

    # -- properties (PySurface.cpp:111-160) ---------------------------------------
    @property
    def Width(self) -> int:
        return self._w

    @property
    def Height(self) -> int:
        return self._h

    @property
    def Pitch(self) -> int:
        return self._pitch

    @property
    def ElemSize(self) -> int:
        return self._elem

    @property
    def HostFrameSize(self) -> int:
        return self._w * self._h * self._elem

    @property
    def GpuMem(self) -> int:
        return self._ptr

    @property
    def DeviceId(self) -> int:
        return self._device

    @property
    def OwnMemory(self) -> bool:
        return self._own

    @property
    def Empty(self) -> bool:
        return self._ptr == 0

    # -- DLPack / array interface ---------------------------------------------------
    def __dlpack_device__(self) -> Tuple[int, int]:
        if self._from_dlpack:
            raise RuntimeError("Cant get __dlpack_device__ attribute from Surface created from DLPack.")
        return (int(DLDeviceType.kDLROCM), self._device)

    def _export(self, shape: Sequence[int], strides: Sequence[int]):
        if self._from_dlpack:
            raise RuntimeError("Cant put DLPack SurfacePlane to DLPack")
        if not self._ptr:
            raise RuntimeError("Empty SurfacePlane")
        return shim.dlpack_export(self._ptr, list(shape), list(strides), self._code,
                                  self._elem * 8, int(DLDeviceType.kDLROCM), self._device, self)

    def __dlpack__(self, stream=None, **_kwargs):
        """(H, W) tensor, row stride = pitch/elem (SurfacePlane.cpp:244-281)."""
        return self._export((self._h, self._w), (self._pitch // self._elem, 1))

    def _cai(self, shape, strides) -> dict:
        from .runtime import HipResMgr

        return {
            "shape": tuple(shape),
            "typestr": self._typestr,
            "data": (self._ptr, False),
            "version": 3,
            "strides": tuple(strides),
            "stream": HipResMgr.Instance().GetStream(self._device),
        }

    @property
    def __cuda_array_interface__(self) -> dict:
        """SurfacePlane.cpp:331-371; 3 slots like the reference (trailing zeros)."""
        return self._cai((self._h, self._w, 0), (self._pitch, self._elem, 0))

    def __repr__(self) -> str:
        return (f"Width:        {self._w}\nHeight:       {self._h}\nPitch:        {self._pitch}\n"
                f"ElemSize:     {self._elem}\nGpuMem:       {hex(self._ptr)}\n")


# ---------------------------------------------------------------------------------
# Format table: replaces the 14 Surface* classes of src/TC/src/Surfaces.cpp.
# ---------------------------------------------------------------------------------
@dataclass(frozen=True)
class FormatSpec:
    elem_size: int
    type_code: int
    typestr: str
    num_planes: int
    num_components: int
    # (W, H) in pixels -> list of (width_in_elements, height) per plane
    plane_geometry: Callable[[int, int], List[Tuple[int, int]]]
    layout: str  # "HW", "HWC", "CHW" or "" (multi-plane: no single tensor view)


def _g_single(w, h):
    return [(w, h)]


def _g_semiplanar(w, h):           # Surfaces.cpp:104-113
    return [(w, h * 3 // 2)]


def _g_420(w, h):                  # Surfaces.cpp:231-246
    return [(w, h), (w // 2, h // 2), (w // 2, h // 2)]


def _g_422(w, h):                  # Surfaces.cpp:325-340
    return [(w, h), (w // 2, h), (w // 2, h)]


def _g_444(w, h):                  # Surfaces.cpp:392-407
    return [(w, h)] * 3


def _g_packed3(w, h):              # Surfaces.cpp:465-473
    return [(w * 3, h)]


def _g_planar3(w, h):              # Surfaces.cpp:576-586
    return [(w, h * 3)]


F = PixelFormat
FORMATS = {
    F.Y: FormatSpec(1, kDLUInt, "<u1", 1, 1, _g_single, "HW"),
    F.NV12: FormatSpec(1, kDLUInt, "<u1", 1, 2, _g_semiplanar, "HW"),
    F.P10: FormatSpec(2, kDLUInt, "<u2", 1, 2, _g_semiplanar, "HW"),
    F.P12: FormatSpec(2, kDLUInt, "<u2", 1, 2, _g_semiplanar, "HW"),
    F.YUV420: FormatSpec(1, kDLUInt, "<u1", 3, 3, _g_420, ""),
    F.YUV420_10bit: FormatSpec(2, kDLUInt, "<u2", 3, 3, _g_420, ""),
    F.YUV422: FormatSpec(1, kDLUInt, "<u1", 3, 3, _g_422, ""),
    F.YUV444: FormatSpec(1, kDLUInt, "<u1", 3, 3, _g_444, ""),
    F.YUV444_10bit: FormatSpec(2, kDLUInt, "<u2", 3, 3, _g_444, ""),
    F.RGB: FormatSpec(1, kDLUInt, "<u1", 1, 1, _g_packed3, "HWC"),
    F.BGR: FormatSpec(1, kDLUInt, "<u1", 1, 1, _g_packed3, "HWC"),
    F.RGB_32F: FormatSpec(4, kDLFloat, "<f4", 1, 1, _g_packed3, "HWC"),
    F.RGB_PLANAR: FormatSpec(1, kDLUInt, "<u1", 1, 3, _g_planar3, "CHW"),
    F.RGB_32F_PLANAR: FormatSpec(4, kDLFloat, "<f4", 1, 3, _g_planar3, "CHW"),
}
_SEMIPLANAR = (F.NV12, F.P10, F.P12)
_PACKED3 = (F.RGB, F.BGR, F.RGB_32F)
_PLANAR3 = (F.RGB_PLANAR, F.RGB_32F_PLANAR)


class Surface:
    """Image in device memory: 1+ SurfacePlanes (PySurface.cpp:233-553)."""

    def __init__(self, fmt: PixelFormat, planes: Optional[List[SurfacePlane]] = None):
        fmt = PixelFormat(fmt)
        if fmt not in FORMATS:
            raise ValueError(f"Unsupported pixel format: {fmt}")
        self._fmt = fmt
        self._spec = FORMATS[fmt]
        self._planes: List[SurfacePlane] = planes if planes is not None else [
            SurfacePlane() for _ in range(self._spec.num_planes)]
        self._desc = None
        self._keepalive = None  # DLPack holder / foreign object whose memory is borrowed

    # -- factories ------------------------------------------------------------------
    @staticmethod
    def Make(format, width=None, height=None, gpu_id=None, context=None) -> "Surface":
        """Surface.Make(format, width, height, gpu_id) | (format, width, height, context=)
        (PySurface.cpp:333-376, MemoryInterfaces.cpp:369-404).  Surface.Make(format) alone
        makes an empty surface (MemoryInterfaces.cpp:336-367)."""
        fmt = PixelFormat(format)
        if fmt not in FORMATS:
            raise ValueError(f"Unsupported pixel format: {fmt}")
        if width is None and height is None:
            return Surface(fmt)
        if gpu_id is None:
            gpu_id = 0 if context is None else int(context)
        from .runtime import HipResMgr

        HipResMgr.Instance()._check(int(gpu_id))
        width, height = int(width), int(height)
        if width <= 0 or height <= 0:
            raise RuntimeError("Surface.Make: width and height must be positive")
        spec = FORMATS[fmt]
        planes = [SurfacePlane.allocate(w, h, spec.elem_size, spec.type_code, spec.typestr,
                                        int(gpu_id)) for (w, h) in spec.plane_geometry(width, height)]
        return Surface(fmt, planes)

    # -- geometry (Surfaces.cpp Width/Height/Pitch per class) -------------------------
    def _width(self, plane: int = 0) -> int:
        p = self._planes[0 if self._fmt in _SEMIPLANAR else plane]
        return p.Width // 3 if self._fmt in _PACKED3 else p.Width

    def _height(self, plane: int = 0) -> int:
        if self._fmt in _SEMIPLANAR:
            h = self._planes[0].Height
            return h * 2 // 3 if plane == 0 else h // 3
        p = self._planes[plane]
        return p.Height // 3 if self._fmt in _PLANAR3 else p.Height

    def _pitch(self, plane: int = 0) -> int:
        return self._planes[0 if self._fmt in _SEMIPLANAR else plane].Pitch

    @property
    def Width(self) -> int:
        return self._width(0)

    @property
    def Height(self) -> int:
        return self._height(0)

    @property
    def Pitch(self) -> int:
        return self._pitch(0)

    @property
    def Format(self) -> PixelFormat:
        return self._fmt

    @property
    def ElemSize(self) -> int:
        return self._spec.elem_size

    @property
    def NumPlanes(self) -> int:
        return self._spec.num_planes

    @property
    def NumComponents(self) -> int:
        return self._spec.num_components

    @property
    def IsEmpty(self) -> bool:
        return all(p.Empty for p in self._planes)

    @property
    def IsOwnMemory(self) -> bool:
        return all(p.OwnMemory for p in self._planes)

    @property
    def HostSize(self) -> int:
        """Bytes of the tightly packed host image (MemoryInterfaces.cpp:445-452)."""
        return sum(p.HostFrameSize for p in self._planes)

    @property
    def DeviceId(self) -> int:
        return self._planes[0].DeviceId

    @property
    def Planes(self) -> tuple:
        """Non-owning plane copies (PySurface.cpp:538-552)."""
        return tuple(p.view() for p in self._planes)

    def PixelPtr(self, component: int = 0) -> int:
        """Device pointer of a COMPONENT (Surfaces.cpp:170-176, 606-612, 72-74 ...)."""
        if self._fmt in _SEMIPLANAR or self._fmt in _PLANAR3:
            if component >= self._spec.num_components:
                raise ValueError("Invalid component number")
            return self._planes[0].GpuMem + component * self.Height * self.Pitch
        return self._planes[component].GpuMem

    @property
    def Shape(self) -> tuple:
        """numpy-like shape (MemoryInterfaces.cpp:461-479): CAI shape for single-plane
        formats, flat element count otherwise."""
        if self._spec.layout:
            return tuple(d for d in self._cai_shape_strides()[0] if d)
        return (self.HostSize // self._spec.elem_size,)

    # -- the C-ABI view ---------------------------------------------------------------
    def desc(self):
        """vali_surface descriptor (cached): component pointers, pitches, size, format."""
        if self._desc is None:
            nc = self._spec.num_components if self._spec.num_planes == 1 else self._spec.num_planes
            ptrs = [self.PixelPtr(c) for c in range(nc)]
            if self._fmt in _SEMIPLANAR or self._fmt in _PLANAR3:
                pitches = [self.Pitch] * nc
            else:
                pitches = [self._pitch(c) for c in range(nc)]
            self._desc = shim.SurfaceDesc(ptrs, pitches, self.Width, self.Height, int(self._fmt))
        return self._desc

    def _update(self, planes: List[SurfacePlane]) -> bool:
        """Surface::Update: borrow foreign planes, refused when owning memory
        (Surfaces.cpp:82-90)."""
        if (not self.IsEmpty and self.IsOwnMemory) or len(planes) < self._spec.num_planes:
            return False
        if any(p.ElemSize != self._spec.elem_size for p in planes[: self._spec.num_planes]):
            return False
        self._planes = [p.view() for p in planes[: self._spec.num_planes]]
        self._desc = None
        return True

    def Clone(self) -> "Surface":
        """Deep copy into fresh memory (MemoryInterfaces.cpp:406-436)."""
        if self.IsEmpty:
            return Surface(self._fmt)
        dev = self.DeviceId
        new = Surface.Make(self._fmt, self.Width, self.Height, dev)
        from .runtime import HipResMgr

        stream = HipResMgr.Instance().GetStream(dev)
        for s, d in zip(self._planes, new._planes):
            shim.memcpy2d_async(dev, d.GpuMem, d.Pitch, s.GpuMem, s.Pitch,
                                s.Width * s.ElemSize, s.Height, 2, stream)
        shim.stream_sync(dev, stream)
        return new

    # -- DLPack / CAI -------------------------------------------------------------------
    def _require_single_plane(self, what: str):
        if self.NumPlanes > 1:
            raise RuntimeError(f"Surface has multiple planes. Use {what} methods for particular "
                               "plane instead.")

    def __dlpack_device__(self):
        self._require_single_plane("DLPack")
        return self._planes[0].__dlpack_device__()

    def __dlpack__(self, stream=None, **_kwargs):
        """Per-format tensor view (Surfaces.cpp:512-542 HWC, 631-661 CHW, 197-203 NV12 as-is)."""
        self._require_single_plane("DLPack")
        p = self._planes[0]
        e = p.ElemSize
        if self._fmt in _PACKED3:
            return p._export((self.Height, self.Width, 3), (p.Pitch // e, 3, 1))
        if self._fmt in _PLANAR3:
            return p._export((3, self.Height, self.Width),
                             (p.Pitch * self.Height // e, p.Pitch // e, 1))
        return p.__dlpack__()

    def _cai_shape_strides(self):
        p = self._planes[0]
        e = p.ElemSize
        if self._fmt in _PACKED3:      # Surfaces.cpp:544-555
            return (self.Height, self.Width, 3), (p.Pitch, e * 3, e)
        if self._fmt in _PLANAR3:      # Surfaces.cpp:663-674
            return (3, self.Height, self.Width), (p.Pitch * self.Height, p.Pitch, e)
        return (p.Height, p.Width, 0), (p.Pitch, e, 0)

    @property
    def __cuda_array_interface__(self) -> dict:
        self._require_single_plane("CAI")
        shape, strides = self._cai_shape_strides()
        return self._planes[0]._cai(shape, strides)

    @staticmethod
    def from_dlpack(capsule, format=PixelFormat.RGB) -> "Surface":
        """Borrow a 2-D device tensor as a single-plane Surface
        (PySurface.cpp:446-474, SurfacePlane.cpp:89-121).  Accepts a DLPack capsule or
        any object with __dlpack__.  kDLROCM (and kDLCUDA, which PyTorch-ROCm builds
        may also report) are accepted."""
        if hasattr(capsule, "__dlpack__") and type(capsule).__name__ != "PyCapsule":
            capsule = capsule.__dlpack__()
        info, holder = shim.dlpack_import(capsule)
        if len(info["shape"]) != 2:
            raise RuntimeError("Only 2D tensors are supported.")
        if info["device_type"] not in (int(DLDeviceType.kDLROCM), int(DLDeviceType.kDLCUDA)):
            raise RuntimeError("Only kDLROCM tensors are supported.")
        if info["lanes"] != 1:
            raise RuntimeError("Only 1 lane tensors are supported.")
        if info["code"] not in (kDLUInt, kDLFloat):
            raise RuntimeError("Only kDLUInt and kDLFloat tensors are supported.")
        if info["strides"][1] != 1:
            raise RuntimeError("Only tensors with contiguous rows are supported.")
        elem = info["bits"] // 8
        typestr = {(kDLUInt, 1): "<u1", (kDLUInt, 2): "<u2", (kDLFloat, 4): "<f4"}.get(
            (info["code"], elem), "<u1")
        plane = SurfacePlane(info["shape"][1], info["shape"][0], info["strides"][0] * elem, elem,
                             info["code"], typestr, info["ptr"], info["device_id"], holder, False,
                             True)
        surf = Surface(PixelFormat(format))
        if not surf._update([plane]):
            raise RuntimeError("Failed to make Surface.")
        surf._keepalive = holder
        return surf

    @staticmethod
    def from_cai(dict, format=PixelFormat.RGB) -> "Surface":  # noqa: A002 -- the reference's keyword
        """Borrow memory described by __cuda_array_interface__ (v3)
        (PySurface.cpp:468-535, SurfacePlane.cpp:123-169).  The first parameter is an OBJECT
        exposing the attribute; its name `dict` is the reference's (py::arg("dict"), :529)."""
        obj = dict
        if not hasattr(obj, "__cuda_array_interface__"):
            raise RuntimeError("'__cuda_array_interface__' not found")
        cai = obj.__cuda_array_interface__
        fmt = PixelFormat(format)
        for key in cai:
            if key not in ("shape", "strides", "typestr", "data", "stream", "version", "descr",
                           "mask"):
                raise RuntimeError("Unsupported attribute " + key)
        if cai.get("version", 3) != 3 and cai.get("version", 3) != 2:
            raise RuntimeError("Unsupported version")
        shape = tuple(int(x) for x in cai["shape"])
        typestr = cai["typestr"]
        if typestr not in ("<u1", "|u1", "<u2", "|u2", "<f4", "|f4"):
            raise RuntimeError("Only u8, u16 and f32 tensors are supported.")
        elem = int(typestr[-1])
        ptr, read_only = cai["data"]
        if read_only:
            raise RuntimeError("Read-only tensors are not supported.")
        if len(shape) < 2:
            raise RuntimeError("Only 2D tensors are supported.")
        strides = cai.get("strides")
        if strides is None:  # C-contiguous
            acc, strides = elem, []
            for d in reversed(shape):
                strides.insert(0, acc)
                acc *= d
        strides = tuple(int(s) for s in strides)
        layout = FORMATS[fmt].layout if fmt in FORMATS else ""
        if layout == "HW":
            h, w, pitch = shape[0], shape[1], strides[0]
        elif layout == "HWC":
            if len(shape) < 3:
                raise RuntimeError("HWC layout needs a 3-D tensor")
            h, w, pitch = shape[0], shape[1] * shape[2], strides[0]
        elif layout == "CHW":
            if len(shape) < 3:
                raise RuntimeError("CHW layout needs a 3-D tensor")
            h, w, pitch = shape[0] * shape[1], shape[2], strides[1]
        else:
            raise RuntimeError("Only HW, HWC and CHW layouts are supported.")
        if not pitch:
            pitch = w * elem
        device = shim.ptr_device(int(ptr))
        code = kDLFloat if typestr[-2] == "f" else kDLUInt
        stream = cai.get("stream")
        if stream is not None and int(stream) not in (0, 1, 2):
            shim.stream_sync(device, int(stream))   # CudaStrSync (SurfacePlane.cpp:168)
        plane = SurfacePlane(w, h, pitch, elem, code, typestr.replace("|", "<"), int(ptr), device,
                             obj, False, False)
        surf = Surface(fmt)
        if not surf._update([plane]):
            raise RuntimeError("Failed to make Surface.")
        surf._keepalive = obj
        return surf

    def __repr__(self) -> str:
        return (f"Width:            {self.Width}\nHeight:           {self.Height}\n"
                f"Format:           {self._fmt.name}\nPitch:            {self.Pitch}\n"
                f"Elem size(bytes): {self.ElemSize}\nNum planes:       {self.NumPlanes}\n")
