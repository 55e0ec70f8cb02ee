"""PySurfaceConverter / PySurfaceResizer / PySurfaceRotator / PySurfaceUD.

Host-side mirror of the reference's L3 tasks + L4 Py* wrappers: validation, format-pair
dispatch, colour-variant selection and error codes are restated here in Python; the
per-pixel work is one C-ABI call into libvali_hip.so per Run
(reference: src/TC/src/TaskConvertSurface.cpp:966-1095, src/python_vali/src/PySurfaceConverter.cpp:26-161).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from ._native import shim
from .enums import (ColorRange, ColorSpace, ColorspaceConversionContext, Interpolation, PixelFormat,
                    TaskExecDetails, TaskExecInfo)
from .runtime import CudaStreamEvent, HipResMgr, is_capturing
from .surface import Surface

F = PixelFormat

# ---- colour matrices ------------------------------------------------------------------
# The constants NVIDIA documents for the NPP functions the reference calls;
# the CSC struct layout is include/vali_hip.h:vali_csc.  (y0, cy, crv, cgu, cgv, cbu)
CSC_NPP_YUV = (0.0, 1.0, 1.140, -0.394, -0.581, 2.032)          # nppiNV12ToRGB / YUV420ToRGB / YUVToRGB
CSC_NPP_709CSC = (16.0, 1.164, 1.793, -0.213, -0.533, 2.112)    # nppiNV12ToRGB_709CSC
CSC_NPP_709HDTV = (0.0, 1.0, 1.5748, -0.1873, -0.4681, 1.8556)  # nppiNV12ToRGB_709HDTV
CSC_NPP_YCBCR = (16.0, 1.164, 1.596, -0.392, -0.813, 2.017)     # nppiYCbCr420ToRGB / YCbCrToBGR

_csc_cache = {}


def _csc(coeffs):
    c = _csc_cache.get(coeffs)
    if c is None:
        c = _csc_cache[coeffs] = shim.Csc(*coeffs)
    return c


def _status(rc: int) -> TaskExecDetails:
    """C status -> TaskExecDetails (the reference maps NppStatus the same way, e.g.
    TaskConvertSurface.cpp:151-155)."""
    if rc == 0:              # shim.OK; the shared immutable success object (hot path)
        return _S_OK
    if rc == shim.ERR_INVALID_ARG:
        return TaskExecDetails.failed(TaskExecInfo.INVALID_INPUT, shim.last_error())
    if rc == shim.ERR_UNSUPPORTED:
        return TaskExecDetails.failed(TaskExecInfo.NOT_SUPPORTED, shim.last_error())
    raise RuntimeError("HIP failure: " + shim.last_error())


_S_OK = TaskExecDetails.ok()
_OK_PAIR = (True, TaskExecInfo.SUCCESS)


def _cc_key(cc_ctx):
    """memo key of a colour context: its VALUES (the object is mutable, as in the reference)"""
    return None if cc_ctx is None else (cc_ctx.color_space, cc_ctx.color_range)
_S_INVALID = TaskExecDetails.failed(TaskExecInfo.INVALID_INPUT, "invalid src / dst")
_S_UNSUPP_CC = TaskExecDetails.failed(TaskExecInfo.UNSUPPORTED_FMT_CONV_PARAMS,
                                      "unsupported cc_ctx params")


def _nv12_variant(cc_ctx: Optional[ColorspaceConversionContext]):
    """Colour-variant switch of nv12_rgb (TaskConvertSurface.cpp:117-149):
    default BT709+JPEG; BT709: JPEG -> 709HDTV else 709CSC; BT601: JPEG -> 'YUV',
    otherwise unsupported; any other colour space unsupported."""
    space = cc_ctx.color_space if cc_ctx else ColorSpace.BT_709
    rng = cc_ctx.color_range if cc_ctx else ColorRange.JPEG
    if space == ColorSpace.BT_709:
        return CSC_NPP_709HDTV if rng == ColorRange.JPEG else CSC_NPP_709CSC
    if space == ColorSpace.BT_601 and rng == ColorRange.JPEG:
        return CSC_NPP_YUV
    return None


_MISSING = object()
_nv12_csc_cache = {}    # (color_space, color_range) | None -> shim.Csc | None (unsupported)


def _nv12_rgb(io, stream: int, cc_ctx, csc=None) -> TaskExecDetails:
    """nv12_rgb / nv12_bgr (TaskConvertSurface.cpp:61-156).  The reference's nv12_bgr
    falls off the end of the function (its return sits after `break`, :100-105); here
    BGR simply follows the RGB logic with the channel order reversed."""
    if csc is None:     # `csc` = a matrix handed in from outside (the multi-GPU broadcast)
        key = (cc_ctx.color_space, cc_ctx.color_range) if cc_ctx is not None else None
        csc = _nv12_csc_cache.get(key, _MISSING)
        if csc is _MISSING:
            coeffs = _nv12_variant(cc_ctx)
            csc = _nv12_csc_cache[key] = _csc(coeffs) if coeffs is not None else None
        if csc is None:
            return _S_UNSUPP_CC
    return _status(io.nv12_to_rgb(stream, csc))


# RGB -> YUV matrices NVIDIA documents for nppiRGBToYUV / nppiRGBToYCbCr;
# rows = (kR, kG, kB, offset) for Y, U/Cb, V/Cr.  Row 0 of the first is nppiRGBToGray.
RGB2YUV_NPP_YUV = ((0.299, 0.587, 0.114, 0.0), (-0.147, -0.289, 0.436, 128.0),
                   (0.615, -0.515, -0.100, 128.0))
RGB2YUV_NPP_YCBCR = ((0.257, 0.504, 0.098, 16.0), (-0.148, -0.291, 0.439, 128.0),
                     (0.439, -0.368, -0.071, 128.0))

_params_cache = {}


def _params(csc=None, rgb2yuv=None):
    key = (csc, rgb2yuv)
    p = _params_cache.get(key)
    if p is None:
        p = _params_cache[key] = shim.CvtParams(_csc(csc) if csc else None,
                                                [list(r) for r in rgb2yuv] if rgb2yuv else [])
    return p


_S_FAIL = TaskExecDetails.failed(TaskExecInfo.FAIL)


def _space_range(cc_ctx, space_default, range_default):
    return ((cc_ctx.color_space if cc_ctx else space_default),
            (cc_ctx.color_range if cc_ctx else range_default))


class _Single:
    """One (src, dst) pair: a C-ABI call per Run."""

    def __init__(self, src, dst):
        self.src, self.dst = src, dst
        self.last = None    # (C-ABI entry, its parameter block) of the call made: the converter's memo

    def convert(self, stream, params):
        self.last = (shim.convert, params)
        return shim.convert(self.src.desc(), self.dst.desc(), params, stream)

    def nv12_to_rgb(self, stream, csc):
        self.last = (shim.nv12_to_rgb, csc)
        return shim.nv12_to_rgb(self.src.desc(), self.dst.desc(), csc, stream)


class _Batched:
    """n pairs of one geometry: ONE launch (device descriptor arrays of a SurfaceBatch)."""

    def __init__(self, batch):
        self.b = batch

    def convert(self, stream, params):
        b = self.b
        return shim.convert_batch(b.d_src, b.d_dst, b.n, int(b.src_format), int(b.dst_format),
                                  b.src_size[0], b.src_size[1], params, stream)

    def nv12_to_rgb(self, stream, csc):
        b = self.b
        return shim.nv12_to_rgb_batch(b.d_src, b.d_dst, b.n, b.src_size[0], b.src_size[1],
                                      int(b.dst_format), csc, stream)


def _convert(io, stream, params) -> TaskExecDetails:
    return _status(io.convert(stream, params))


def _plain(io, stream, cc_ctx, csc=None):
    """Pairs without colour parameters: (de)interleave, swap, copies, element types."""
    return _convert(io, stream, _params())


def _nv12_yuv420(io, stream, cc_ctx, csc=None):
    # nv12_yuv420 (TaskConvertSurface.cpp:158-200): JPEG and MPEG pick two NPP entry points
    # that do the same byte shuffle; any other range is refused.
    _, rng = _space_range(cc_ctx, ColorSpace.BT_601, ColorRange.JPEG)
    if rng not in (ColorRange.JPEG, ColorRange.MPEG):
        return _S_UNSUPP_CC
    return _convert(io, stream, _params())


def _yuv420_rgb(io, stream, cc_ctx, csc=None):
    # yuv420_rgb / yuv420_bgr (:254-344): BT.601 only; JPEG -> "YUV", otherwise YCbCr
    space, rng = _space_range(cc_ctx, ColorSpace.BT_601, ColorRange.JPEG)
    if space != ColorSpace.BT_601:
        return _S_UNSUPP_CC
    return _convert(io, stream,
                    _params(csc=CSC_NPP_YUV if rng == ColorRange.JPEG else CSC_NPP_YCBCR))


def _yuv444_bgr(io, stream, cc_ctx, csc=None):
    # yuv444_bgr (:346-391): MPEG -> YCbCr, JPEG -> YUV, else NPP_NO_OPERATION_WARNING -> FAIL
    space, rng = _space_range(cc_ctx, ColorSpace.BT_601, ColorRange.JPEG)
    if space != ColorSpace.BT_601:
        return _S_UNSUPP_CC
    if rng == ColorRange.MPEG:
        return _convert(io, stream, _params(csc=CSC_NPP_YCBCR))
    if rng == ColorRange.JPEG:
        return _convert(io, stream, _params(csc=CSC_NPP_YUV))
    return _S_FAIL


def _yuv444_rgb(io, stream, cc_ctx, csc=None):
    # yuv444_rgb (:393-434): JPEG only
    space, rng = _space_range(cc_ctx, ColorSpace.BT_601, ColorRange.JPEG)
    if space != ColorSpace.BT_601:
        return _S_UNSUPP_CC
    if rng != ColorRange.JPEG:
        return _S_FAIL
    return _convert(io, stream, _params(csc=CSC_NPP_YUV))


def _rgb_to_yuv(io, stream, cc_ctx, csc=None):
    # bgr_yuv444 / rgb_yuv444 / rgb_planar_yuv444 / rgb_yuv420 (:481-619, :657-704):
    # BT.601 only; JPEG -> nppiRGBToYUV*, MPEG -> nppiRGBToYCbCr*, else FAIL.
    # (The reference's rgb_yuv444 + MPEG writes PACKED YCbCr into plane 0, :557-560 -- a bug;
    # here it produces planar YCbCr like the other variants.)
    space, rng = _space_range(cc_ctx, ColorSpace.BT_601, ColorRange.JPEG)
    if space != ColorSpace.BT_601:
        return _S_UNSUPP_CC
    if rng == ColorRange.JPEG:
        return _convert(io, stream, _params(rgb2yuv=RGB2YUV_NPP_YUV))
    if rng == ColorRange.MPEG:
        return _convert(io, stream, _params(rgb2yuv=RGB2YUV_NPP_YCBCR))
    return _S_FAIL


def _rgb_y(io, stream, cc_ctx, csc=None):
    # rbg8_y (:232-252): nppiRGBToGray
    return _convert(io, stream, _params(rgb2yuv=RGB2YUV_NPP_YUV))


# (src, dst) -> implementation; order is GetSupportedConversions()
# (TaskConvertSurface.cpp:966-994).  NV12 -> RGB_PLANAR is an extension: the fused form
# of the reference's NV12->RGB->RGB_PLANAR chain (BASELINE config 2).
_CONVERSIONS = {
    (F.NV12, F.YUV420): _nv12_yuv420,
    (F.YUV420, F.NV12): _plain,
    (F.P10, F.NV12): _plain,
    (F.P12, F.NV12): _plain,
    (F.NV12, F.RGB): _nv12_rgb,
    (F.NV12, F.BGR): _nv12_rgb,
    (F.RGB, F.RGB_PLANAR): _plain,
    (F.RGB_PLANAR, F.RGB): _plain,
    (F.RGB_PLANAR, F.YUV444): _rgb_to_yuv,
    (F.Y, F.YUV444): _plain,
    (F.YUV420, F.RGB): _yuv420_rgb,
    (F.RGB, F.YUV420): _rgb_to_yuv,
    (F.RGB, F.YUV444): _rgb_to_yuv,
    (F.RGB, F.BGR): _plain,
    (F.BGR, F.RGB): _plain,
    (F.YUV420, F.BGR): _yuv420_rgb,
    (F.YUV444, F.BGR): _yuv444_bgr,
    (F.YUV444, F.RGB): _yuv444_rgb,
    (F.BGR, F.YUV444): _rgb_to_yuv,
    (F.NV12, F.Y): _plain,
    (F.RGB, F.RGB_32F): _plain,
    (F.RGB, F.Y): _rgb_y,
    (F.RGB_32F, F.RGB_32F_PLANAR): _plain,
    (F.NV12, F.RGB_PLANAR): _nv12_rgb,
}


class _SurfaceTask:
    def __init__(self, gpu_id: int, stream=None):
        self._gpu_id = int(gpu_id)
        # stream=None: this GPU's stream from the resource manager (the reference's one-argument constructors,
        # CudaResMgr::GetStream).  An explicit stream is used as handed over (PySurfaceConverter.cpp:33-45) -- also 0,
        # the legacy default stream: torch.cuda.current_stream().cuda_stream is 0 for torch's default stream, and a
        # caller who passes it expects the task to be ordered with their own work there.  A null hipStream_t carries no
        # device, so such a task makes its GPU current on the calling thread before every call (_bind_null_stream).
        self._stream = HipResMgr.Instance().GetStream(self._gpu_id) if stream is None else int(stream)
        if self._stream == 0:
            self._bind_null_stream()
        self._event = CudaStreamEvent(self._stream, self._gpu_id)
        # Memo of recent successful single-surface calls: (src descriptor, dst descriptor, extra
        # key...) -> (C-ABI entry, arguments between the descriptors and the stream).
        # A repeated RunAsync on the same surfaces (the per-frame loop of every sample pipeline,
        # one entry per step of its chain; BASELINE config 2) then costs one C call instead of the
        # task's Python dispatch: 1 - 2.5 us less per call (4.95 -> 3.9 us for the converter, the
        # host's own launch rate).  The memo keeps the (pointer-only) descriptors alive, not the
        # surfaces; a Surface gets a NEW descriptor object when it is re-pointed (Surface._update)
        # and a new Surface has its own, so a stale entry can never match.
        self._memo = {}
        self._batches = {}    # (src descriptors, dst descriptors) -> SurfaceBatch, see _batch_of

    def _bind_null_stream(self):
        """Tasks on the null stream: a null hipStream_t carries no device, so every Run* / PrepareBatch makes the task's
        GPU the thread's current device for the duration of the call and puts the caller's device BACK afterwards (the
        reference's CudaCtxPush / pop, CudaUtils.hpp:77-90): a host application that runs a task on (gpu 1, stream 0)
        keeps allocating on the device it had selected.  Done by wrapping the methods of THIS instance, so tasks on real
        streams -- whose device the library reads off the stream -- keep their call path untouched; the wrappers hold a
        weak reference to the task (no self-referencing cycle: the task's events and memo go when its last user does)."""
        import inspect
        import weakref

        gpu, ref = self._gpu_id, weakref.ref(self)
        for name in dir(type(self)):
            if name.startswith("Run") or name == "PrepareBatch":
                # (getattr_static: the raw class attribute -- getattr resolves descriptors, so a static Run* helper would
                # look like a plain function here and be called with an extra `me`)
                if isinstance(inspect.getattr_static(type(self), name), (staticmethod, classmethod)):
                    continue
                unbound = getattr(type(self), name)
                if not callable(unbound):
                    continue

                def call(*a, _fn=unbound, _name=name, **k):
                    me = ref()
                    if me is None:     # a bound method kept (`f = task.Run`) after the task itself was dropped
                        raise ReferenceError(f"{_name}: the task this method belonged to no longer exists")
                    prev = shim.device_get()
                    if prev != gpu:
                        shim.device_set(gpu)
                    try:
                        return _fn(me, *a, **k)
                    finally:
                        if prev != gpu and prev >= 0:
                            shim.device_set(prev)
                setattr(self, name, call)

    def _batch_of(self, batch, dsts):
        """RunBatch*(srcs, dsts): the descriptor arrays of a (srcs, dsts) pair of lists are uploaded ONCE and
        kept with the task (8 most recent pairs) -- a repeated call costs no allocation, no copy and no
        synchronisation, and the arrays outlive every launch that reads them.  RunBatch*(batch) passes a
        prepared SurfaceBatch through."""
        if isinstance(batch, SurfaceBatch):
            return batch
        if dsts is None:
            raise ValueError("RunBatch: pass a SurfaceBatch, or two lists (srcs, dsts)")
        srcs, dsts = list(batch), list(dsts)
        key = (tuple(s.desc() for s in srcs), tuple(d.desc() for d in dsts))
        b = self._batches.pop(key, None)
        if b is None:
            # a miss while the stream is capturing must fail BEFORE anything is evicted: dropping a cached batch
            # synchronises its stream (SurfaceBatch.__del__), which would invalidate the capture instead of reporting it
            if is_capturing(self._gpu_id, self._stream):
                raise RuntimeError("RunBatch(srcs, dsts): these lists have no prepared batch and one cannot be created "
                                   "while the stream is capturing -- call PrepareBatch() before the capture")
            b = SurfaceBatch(self._gpu_id, self._stream, srcs, dsts)
            # the cache must not pin the caller's surfaces (a task fed fresh lists of 512 2160p frames would hold
            # 8 x 19 GB): it keeps the descriptor arrays only; a hit needs the SAME descriptor objects, i.e. live
            # surfaces that were not re-pointed, and a freed surface's descriptors can never match a new one's
            b._keep = None
            while len(self._batches) >= 8:                     # least recently used first (a hit re-inserts below)
                self._batches.pop(next(iter(self._batches)))
        self._batches[key] = b
        return b

    def _memo_put(self, key, fn, args, keep=None):
        if len(self._memo) >= 16:
            self._memo.clear()
        self._memo[key] = (fn, args, keep)

    @property
    def Stream(self) -> int:
        return self._stream

    def _sync(self):
        # the blocking Run* forms: everything issued on the task's stream has finished.  vali_stream_wait = a completion
        # word the stream writes into pinned host memory, spun on with the GIL released: 8.7 us around a 1 us kernel
        # against 11.8-12.6 us for hipStreamSynchronize (an event record + hipEventSynchronize: the same or worse;
        # interrupts off / ROC_ACTIVE_WAIT_TIMEOUT: no change) -- tools/exp/sync_latency.hip; the launch -> completion
        # round trip of a kernel that announces its own end is 8.15 us on the same box, so that is the floor.
        shim.stream_wait(self._gpu_id, self._stream)


class PySurfaceConverter(_SurfaceTask):
    """GPU colour-space / pixel-format conversion.

    reference: src/python_vali/src/PySurfaceConverter.cpp:26-161 (ctor, Run, RunAsync,
    Conversions, Stream); dispatch ConvertSurface::Run (TaskConvertSurface.cpp:1009-1095).
    RunBatch is new: one launch over a list of same-shaped surfaces.
    """

    def __init__(self, gpu_id: int, stream=None):
        super().__init__(gpu_id, stream)
        self._batch_cache = {}

    @staticmethod
    def Conversions() -> List[Tuple[PixelFormat, PixelFormat]]:
        return list(_CONVERSIONS.keys())

    def _run(self, src: Surface, dst: Surface, cc_ctx) -> TaskExecDetails:
        if src.Width != dst.Width or src.Height != dst.Height:   # Validate(), :1001-1015
            return _S_INVALID
        impl = _CONVERSIONS.get((src.Format, dst.Format))
        if impl is None:                                          # :1085-1089
            raise ValueError(f"Unsupported pixel format conversion: {src.Format.name} -> "
                             f"{dst.Format.name}")
        io = _Single(src, dst)
        d = impl(io, self._stream, cc_ctx)
        if d is _S_OK and io.last is not None:
            self._memo_put((src.desc(), dst.desc(), _cc_key(cc_ctx)), io.last[0], (io.last[1],))
        return d

    def RunAsync(self, src: Surface, dst: Surface,
                 cc_ctx: Optional[ColorspaceConversionContext] = None) -> Tuple[bool, TaskExecInfo]:
        try:
            d1, d2 = src._desc, dst._desc
            m = self._memo.get((d1, d2, None if cc_ctx is None else (cc_ctx.color_space, cc_ctx.color_range)))
        except (AttributeError, TypeError):     # not Surfaces: let the dispatch below complain
            m = None
        if m is not None and m[0](d1, d2, *m[1], self._stream) == 0:
            return _OK_PAIR
        d = self._run(src, dst, cc_ctx)
        return d.success, d.info

    def Run(self, src: Surface, dst: Surface,
            cc_ctx: Optional[ColorspaceConversionContext] = None) -> Tuple[bool, TaskExecInfo]:
        r = self.RunAsync(src, dst, cc_ctx)     # (the memoised C call when the pair repeats)
        self._sync()
        return r

    # -- batched form ------------------------------------------------------------------
    def PrepareBatch(self, srcs: Sequence[Surface], dsts: Sequence[Surface]) -> "SurfaceBatch":
        """Upload the descriptor arrays of a (srcs, dsts) batch once; reuse with RunBatch."""
        return SurfaceBatch(self._gpu_id, self._stream, srcs, dsts)

    def RunBatchAsync(self, batch, dsts=None, cc_ctx=None, csc=None) -> Tuple[bool, TaskExecInfo]:
        """RunBatchAsync(batch, cc_ctx=...) or RunBatchAsync(srcs, dsts, cc_ctx).
        All-or-nothing, no per-item sync (idiom of PyNvJpegEncoder.Run,
        src/python_vali/src/PyNvJpegEncoder.cpp:31-81).  `csc` overrides the matrix
        cc_ctx would select (the multi-GPU pipeline passes the broadcast block)."""
        batch = self._batch_of(batch, dsts)
        impl = _CONVERSIONS.get((batch.src_format, batch.dst_format))
        if impl is None:
            raise ValueError(f"Unsupported pixel format conversion: {batch.src_format.name} -> "
                             f"{batch.dst_format.name}")
        if batch.src_size != batch.dst_size:
            return False, TaskExecInfo.INVALID_INPUT
        d = impl(_Batched(batch), self._stream, cc_ctx, csc)
        return d.success, d.info

    def RunBatch(self, batch, dsts=None, cc_ctx=None, csc=None) -> Tuple[bool, TaskExecInfo]:
        r = self.RunBatchAsync(batch, dsts, cc_ctx, csc)
        self._sync()
        return r


class SurfaceBatch:
    """Device-resident descriptor arrays for n (src, dst) surface pairs of one geometry."""

    def __init__(self, gpu_id: int, stream: int, srcs: Sequence[Surface], dsts: Sequence[Surface]):
        srcs, dsts = list(srcs), list(dsts)
        if not srcs or len(srcs) != len(dsts):
            raise ValueError("SurfaceBatch: need equally long, non-empty src and dst lists")
        for group in (srcs, dsts):
            f0, s0 = group[0].Format, (group[0].Width, group[0].Height)
            for s in group:
                if s.Format != f0 or (s.Width, s.Height) != s0 or s.IsEmpty:
                    raise ValueError("SurfaceBatch: surfaces of a batch must share format and size")
        if is_capturing(gpu_id, stream):
            # the upload allocates and synchronises: both are illegal while the stream records a graph, and the
            # recorded launches would keep reading arrays that die with this object
            raise RuntimeError("SurfaceBatch: cannot be created while the stream is capturing -- PrepareBatch() "
                               "before the StreamCapture block and Keep() the batch with the capture")
        self.gpu_id = gpu_id
        self._stream = stream
        self.n = len(srcs)
        self.src_format, self.dst_format = srcs[0].Format, dsts[0].Format
        self.src_size = (srcs[0].Width, srcs[0].Height)
        self.dst_size = (dsts[0].Width, dsts[0].Height)
        self._keep = (srcs, dsts)        # a prepared batch keeps its surfaces alive (the task's own cache does not)
        self.src_components, self.src_planes = srcs[0].NumComponents, srcs[0].NumPlanes
        self.d_src = shim.descs_upload(gpu_id, [s.desc() for s in srcs], stream)
        self.d_dst = shim.descs_upload(gpu_id, [s.desc() for s in dsts], stream)

    def __len__(self):
        return self.n

    def __del__(self):
        if getattr(self, "d_src", 0) or getattr(self, "d_dst", 0):
            try:    # a launch issued on the batch's stream may still be reading the arrays
                shim.stream_sync(self.gpu_id, self._stream)
            except Exception:
                pass
        for name in ("d_src", "d_dst"):
            p = getattr(self, name, 0)
            if p:
                try:
                    shim.mem_free(self.gpu_id, p)
                except Exception:
                    pass
                setattr(self, name, 0)


# ---- PySurfaceUD -------------------------------------------------------------------------
# UDSurface::SupportedConversions() (src/TC/src/UDSurface.cpp:117-133).  The planar-source
# rows (YUV420 -> YUV444, YUV420_10bit -> YUV444_10bit) go through NPP Lanczos in the
# reference (UDPlanar, :33-93): here vali_ud_planar, all three planes in one launch, Lanczos-3.
_UD_CONVERSIONS = [
    (F.NV12, F.YUV444), (F.NV12, F.RGB), (F.NV12, F.RGB_32F), (F.NV12, F.RGB_PLANAR),
    (F.NV12, F.RGB_32F_PLANAR), (F.YUV420, F.YUV444), (F.P10, F.YUV444_10bit), (F.P10, F.RGB_32F),
    (F.P10, F.RGB_32F_PLANAR), (F.YUV420_10bit, F.YUV444_10bit),
]
_UD_SEMIPLANAR = {p for p in _UD_CONVERSIONS if p[0] in (F.NV12, F.P10)}


class PySurfaceUD(_SurfaceTask):
    """Chroma upsample + resize (+ YUV->RGB) in one pass.

    reference: src/python_vali/src/PySurfaceUD.cpp:26-143 (ctor, Run, RunAsync,
    SupportedFormats, Stream); UDSurface::Run (src/TC/src/UDSurface.cpp:135-177).
    As in the reference the dst size alone defines the scale (no size validation).
    """

    @staticmethod
    def SupportedFormats() -> List[Tuple[PixelFormat, PixelFormat]]:
        return list(_UD_CONVERSIONS)

    def _run(self, src: Surface, dst: Surface) -> TaskExecDetails:
        pair = (src.Format, dst.Format)
        if pair not in _UD_CONVERSIONS:                       # UDSurface.cpp:137-149
            return TaskExecDetails.failed(TaskExecInfo.NOT_SUPPORTED)
        if pair not in _UD_SEMIPLANAR:
            # UDPlanar (UDSurface.cpp:33-93): every source plane resized to the matching destination plane
            # with NPPI_INTER_LANCZOS -- one launch over the three planes
            d = _status(shim.ud_planar(src.desc(), dst.desc(), shim.INTERP_LANCZOS, self._stream))
            if d is _S_OK:
                self._memo_put((src.desc(), dst.desc()), shim.ud_planar, (shim.INTERP_LANCZOS,))
            return d
        d = _status(shim.ud_nv12(src.desc(), dst.desc(), self._stream))
        if d is _S_OK:
            self._memo_put((src.desc(), dst.desc()), shim.ud_nv12, ())
        return d

    def RunAsync(self, src: Surface, dst: Surface) -> Tuple[bool, TaskExecInfo]:
        try:
            d1, d2 = src._desc, dst._desc
            m = self._memo.get((d1, d2))
        except (AttributeError, TypeError):
            m = None
        if m is not None and m[0](d1, d2, *m[1], self._stream) == 0:
            return _OK_PAIR
        d = self._run(src, dst)
        return d.success, d.info

    def Run(self, src: Surface, dst: Surface) -> Tuple[bool, TaskExecInfo]:
        r = self.RunAsync(src, dst)
        self._sync()
        return r

    # -- fused UD + quarter-turn rotation (new; BASELINE config 4 as ONE pass) ----------------
    @staticmethod
    def _quarter(angle: float):
        """90 / 180 / 270 (any sign / multiple of 360) -> 1 / 2 / 3; None otherwise."""
        q = float(angle) / 90.0
        return int(round(q)) % 4 if abs(q - round(q)) < 1e-9 and int(round(q)) % 4 else None

    def RunRotatedAsync(self, src: Surface, dst: Surface, angle: float) -> Tuple[bool, TaskExecInfo]:
        """UD with the result written rotated by `angle` (a non-zero multiple of 90 degrees):
        bit-identical to `Run(src, tmp)` + `PySurfaceRotator.Run(tmp, dst, angle)` without the
        intermediate surface.  NV12 -> RGB only; for 90 / 270 `dst` is (UD height) x (UD width)."""
        q = self._quarter(angle)
        if q is None or (src.Format, dst.Format) != (F.NV12, F.RGB):
            return False, TaskExecInfo.NOT_SUPPORTED
        d = _status(shim.ud_nv12_rot(src.desc(), dst.desc(), q, self._stream))
        return d.success, d.info

    def RunRotated(self, src: Surface, dst: Surface, angle: float) -> Tuple[bool, TaskExecInfo]:
        r = self.RunRotatedAsync(src, dst, angle)
        self._sync()
        return r

    def RunRotatedBatchAsync(self, batch, dsts=None, angle: float = 90.0) -> Tuple[bool, TaskExecInfo]:
        batch = self._batch_of(batch, dsts)
        q = self._quarter(angle)
        if q is None or (batch.src_format, batch.dst_format) != (F.NV12, F.RGB):
            return False, TaskExecInfo.NOT_SUPPORTED
        if (batch.src_size[0] | batch.src_size[1]) & 1:       # the one rule for 4:2:0 sizes (include/vali_hip.h)
            return False, TaskExecInfo.INVALID_INPUT
        d = _status(shim.ud_nv12_rot_batch(batch.d_src, batch.d_dst, batch.n, int(batch.src_format),
                                           batch.src_size[0], batch.src_size[1], batch.dst_size[0], batch.dst_size[1],
                                           int(batch.dst_format), q, self._stream))
        return d.success, d.info

    def RunRotatedBatch(self, batch, dsts=None, angle: float = 90.0) -> Tuple[bool, TaskExecInfo]:
        r = self.RunRotatedBatchAsync(batch, dsts, angle)
        self._sync()
        return r

    def RunBatchAsync(self, batch, dsts=None) -> Tuple[bool, TaskExecInfo]:
        batch = self._batch_of(batch, dsts)
        if (batch.src_format, batch.dst_format) not in _UD_CONVERSIONS:
            return False, TaskExecInfo.NOT_SUPPORTED
        if (batch.src_size[0] | batch.src_size[1]) & 1:       # every UD source is 4:2:0: even sizes (include/vali_hip.h)
            return False, TaskExecInfo.INVALID_INPUT
        if (batch.src_format, batch.dst_format) not in _UD_SEMIPLANAR:
            d = _status(shim.ud_planar_batch(batch.d_src, batch.d_dst, batch.n, int(batch.src_format),
                                             int(batch.dst_format), batch.src_size[0], batch.src_size[1],
                                             batch.dst_size[0], batch.dst_size[1], shim.INTERP_LANCZOS, self._stream))
            return d.success, d.info
        d = _status(shim.ud_nv12_batch(batch.d_src, batch.d_dst, batch.n, int(batch.src_format),
                                       batch.src_size[0], batch.src_size[1], batch.dst_size[0], batch.dst_size[1],
                                       int(batch.dst_format), self._stream))
        return d.success, d.info

    def RunBatch(self, batch, dsts=None) -> Tuple[bool, TaskExecInfo]:
        r = self.RunBatchAsync(batch, dsts)
        self._sync()
        return r

    def PrepareBatch(self, srcs: Sequence[Surface], dsts: Sequence[Surface]) -> "SurfaceBatch":
        return SurfaceBatch(self._gpu_id, self._stream, srcs, dsts)


# ---- PySurfacePreprocessor (new: fused inference pre-processing) -------------
class PySurfacePreprocessor(_SurfaceTask):
    """NV12 -> (bilinear resize to dst size) -> RGB -> float -> normalised, ONE launch.

    The reference has no such task: its samples and tests/test_TorchSegmentation.py:176-240
    chain PySurfaceConverter NV12->RGB, RGB->RGB_32F, RGB_32F->RGB_32F_PLANAR and then run
    `torch.divide(x, 255.0)` + torchvision `Normalize(mean, std)` (optionally after a
    PySurfaceResizer).  This task is DEFINED as that chain and is bit-identical to running
    it with this library's own tasks (tests/test_gpu_preproc.py):

        out[c] = ((rgb_u8[c] / 255.0) / div - mean[c]) / std[c]          float32, IEEE

    `div=1, mean=0, std=1` is plain NV12 -> RGB_32F[_PLANAR].  Colour variant selection is
    nv12_rgb's (TaskConvertSurface.cpp:117-149).  dst: RGB_32F_PLANAR or RGB_32F.
    8-bit destinations (RGB, BGR, RGB_PLANAR) give the fused PySurfaceResizer ->
    PySurfaceConverter chain (resize + colour conversion, no float stage); they require the
    identity normalisation.
    """

    def __init__(self, gpu_id: int, stream=None, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0),
                 div: float = 1.0):
        super().__init__(gpu_id, stream)
        self._norm = (float(div), tuple(float(v) for v in mean), tuple(float(v) for v in std))
        if len(self._norm[1]) != 3 or len(self._norm[2]) != 3:
            raise ValueError("mean / std need 3 values (R, G, B)")
        if self._norm[0] == 0.0 or any(v == 0.0 for v in self._norm[2]):
            raise ValueError("div / std must be non-zero")
        self._params_cache = {}

    def _params(self, cc_ctx):
        coeffs = _nv12_variant(cc_ctx)
        if coeffs is None:
            return None
        p = self._params_cache.get(coeffs)
        if p is None:
            div, mean, std = self._norm
            p = self._params_cache[coeffs] = shim.PreprocParams(_csc(coeffs), div, list(mean), list(std))
        return p

    @staticmethod
    def _check(src_fmt, dst_fmt, sw, sh, dw, dh):
        if src_fmt != F.NV12 or dst_fmt not in (F.RGB_32F, F.RGB_32F_PLANAR, F.RGB, F.BGR, F.RGB_PLANAR):
            return TaskExecDetails.failed(TaskExecInfo.NOT_SUPPORTED)
        if (sw | sh | dw | dh) & 1:
            return _S_INVALID
        return None

    def _u8_needs_identity(self, dst_fmt) -> bool:
        """8-bit destinations carry no float stage: a non-identity normalisation cannot apply."""
        return dst_fmt in (F.RGB, F.BGR, F.RGB_PLANAR) and self._norm != (1.0, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))

    def _run(self, src: Surface, dst: Surface, cc_ctx) -> TaskExecDetails:
        if src is None or dst is None or src.IsEmpty or dst.IsEmpty:
            return _S_INVALID
        bad = self._check(src.Format, dst.Format, src.Width, src.Height, dst.Width, dst.Height)
        if bad:
            return bad
        if self._u8_needs_identity(dst.Format):
            return TaskExecDetails.failed(TaskExecInfo.NOT_SUPPORTED)
        p = self._params(cc_ctx)
        if p is None:
            return _S_UNSUPP_CC
        d = _status(shim.nv12_preproc(src.desc(), dst.desc(), p, self._stream))
        if d is _S_OK:
            self._memo_put((src.desc(), dst.desc(), _cc_key(cc_ctx)), shim.nv12_preproc, (p,))
        return d

    def RunAsync(self, src: Surface, dst: Surface, cc_ctx=None) -> Tuple[bool, TaskExecInfo]:
        try:
            d1, d2 = src._desc, dst._desc
            m = self._memo.get((d1, d2, None if cc_ctx is None else (cc_ctx.color_space, cc_ctx.color_range)))
        except (AttributeError, TypeError):
            m = None
        if m is not None and m[0](d1, d2, *m[1], self._stream) == 0:
            return _OK_PAIR
        d = self._run(src, dst, cc_ctx)
        return d.success, d.info

    def Run(self, src: Surface, dst: Surface, cc_ctx=None) -> Tuple[bool, TaskExecInfo]:
        r = self.RunAsync(src, dst, cc_ctx)
        self._sync()
        return r

    def PrepareBatch(self, srcs: Sequence[Surface], dsts: Sequence[Surface]) -> "SurfaceBatch":
        return SurfaceBatch(self._gpu_id, self._stream, srcs, dsts)

    def RunBatchAsync(self, batch, dsts=None, cc_ctx=None) -> Tuple[bool, TaskExecInfo]:
        batch = self._batch_of(batch, dsts)
        bad = self._check(batch.src_format, batch.dst_format, *batch.src_size, *batch.dst_size)
        if bad:
            return bad.success, bad.info
        if self._u8_needs_identity(batch.dst_format):
            return False, TaskExecInfo.NOT_SUPPORTED
        p = self._params(cc_ctx)
        if p is None:
            return False, TaskExecInfo.UNSUPPORTED_FMT_CONV_PARAMS
        d = _status(shim.nv12_preproc_batch(batch.d_src, batch.d_dst, batch.n, batch.src_size[0],
                                            batch.src_size[1], batch.dst_size[0], batch.dst_size[1],
                                            int(batch.dst_format), p, self._stream))
        return d.success, d.info

    def RunBatch(self, batch, dsts=None, cc_ctx=None) -> Tuple[bool, TaskExecInfo]:
        r = self.RunBatchAsync(batch, dsts, cc_ctx)
        self._sync()
        return r


# ---- PySurfaceRotator --------------------------------------------------------------------
_ROT_FORMATS = [F.Y, F.GRAY12, F.RGB, F.BGR, F.RGB_PLANAR, F.YUV420, F.YUV422, F.YUV444, F.RGB_32F,
                F.RGB_32F_PLANAR, F.YUV444_10bit, F.YUV420_10bit]
# format -> (driver, element size); RotateSurface::Run switch (RotateSurface.cpp:168-208)
_ROT_IMPL = {
    F.Y: ("planar", 1), F.RGB: ("packed", 1), F.BGR: ("packed", 1), F.RGB_PLANAR: ("planar", 1),
    F.YUV420: ("planar", 1), F.YUV422: ("planar", 1), F.YUV444: ("planar", 1),
    F.RGB_32F: ("packed", 4), F.RGB_32F_PLANAR: ("planar", 4), F.YUV444_10bit: ("planar", 2),
    F.YUV420_10bit: ("planar", 2),
}


class PySurfaceRotator(_SurfaceTask):
    """Rotate a Surface by an arbitrary angle (bilinear) with optional shift.

    reference: src/python_vali/src/PySurfaceRotator.cpp:26-200; RotateSurface::Run
    (src/TC/src/RotateSurface.cpp:161-214).  Multiples of 90 degrees with zero shifts are
    normalised so the image lands exactly inside the (transposed) destination.
    Deviation: for that special case the reference hands the LUMA-sized shifts to every
    plane, which pushes the half-size chroma planes of YUV420/422 out of their
    destination; here each plane gets the shifts derived from its own size.
    """

    def __init__(self, gpu_id: int, stream=None):
        super().__init__(gpu_id, stream)

    @property
    def SupportedFormats(self) -> List[PixelFormat]:
        return list(_ROT_FORMATS)

    @staticmethod
    def _normalise(angle: float, shift_x: float, shift_y: float):
        """PySurfaceRotator::Run (:40-73): multiples of 90 degrees with zero shifts become
        (normalised angle, per-plane canonical shifts)."""
        import math

        if math.fmod(angle, 90.0) == 0.0 and shift_x == 0.0 and shift_y == 0.0:
            return float((int(round(angle)) + 360) % 360), True
        return float(angle), False

    @staticmethod
    def _check(src_format, dst_format, num_components, num_planes) -> Optional[TaskExecDetails]:
        if src_format != dst_format:                                     # RotateSurface.cpp:162-164
            return TaskExecDetails.failed(TaskExecInfo.SRC_DST_FMT_MISMATCH)
        impl = _ROT_IMPL.get(src_format)
        if impl is None:                                                 # :204-206
            return TaskExecDetails.failed(TaskExecInfo.NOT_SUPPORTED)
        kind, _ = impl
        if kind == "planar" and num_components != num_planes:            # RotPlanar, :135-136
            return TaskExecDetails.failed(TaskExecInfo.INVALID_INPUT)
        if kind == "packed" and num_planes != 1:                         # RotPacked, :152-153
            return TaskExecDetails.failed(TaskExecInfo.INVALID_INPUT)
        return None

    def _run(self, src: Surface, dst: Surface, angle: float, shift_x: float, shift_y: float
             ) -> TaskExecDetails:
        err = self._check(src.Format, dst.Format, src.NumComponents, src.NumPlanes)
        if err is not None:
            return err
        key = (src.desc(), dst.desc(), angle, shift_x, shift_y)
        angle, per_plane = self._normalise(angle, shift_x, shift_y)
        d = _status(shim.rotate(src.desc(), dst.desc(), angle, shift_x, shift_y, int(per_plane),
                                self._stream))
        if d is _S_OK:
            self._memo_put(key, shim.rotate, (angle, shift_x, shift_y, int(per_plane)))
        return d

    def PrepareBatch(self, srcs: Sequence[Surface], dsts: Sequence[Surface]) -> "SurfaceBatch":
        return SurfaceBatch(self._gpu_id, self._stream, srcs, dsts)

    def RunBatchAsync(self, batch, dsts=None, angle: float = 0.0, shift_x: float = 0.0,
                      shift_y: float = 0.0) -> Tuple[bool, TaskExecInfo]:
        """One launch rotates every plane of every surface of the batch."""
        batch = self._batch_of(batch, dsts)
        err = self._check(batch.src_format, batch.dst_format, batch.src_components, batch.src_planes)
        if err is not None:
            return False, err.info
        angle, per_plane = self._normalise(float(angle), float(shift_x), float(shift_y))
        d = _status(shim.rotate_batch(batch.d_src, batch.d_dst, batch.n, int(batch.src_format),
                                      batch.src_size[0], batch.src_size[1], batch.dst_size[0],
                                      batch.dst_size[1], angle, float(shift_x), float(shift_y),
                                      int(per_plane), self._stream))
        return d.success, d.info

    def RunBatch(self, batch, dsts=None, angle: float = 0.0, shift_x: float = 0.0,
                 shift_y: float = 0.0) -> Tuple[bool, TaskExecInfo]:
        r = self.RunBatchAsync(batch, dsts, angle, shift_x, shift_y)
        self._sync()
        return r

    def RunAsync(self, src: Surface, dst: Surface, angle: float, shift_x: float = 0.0,
                 shift_y: float = 0.0) -> Tuple[bool, TaskExecInfo]:
        angle, shift_x, shift_y = float(angle), float(shift_x), float(shift_y)
        try:
            d1, d2 = src._desc, dst._desc
            m = self._memo.get((d1, d2, angle, shift_x, shift_y))
        except (AttributeError, TypeError):
            m = None
        if m is not None and m[0](d1, d2, *m[1], self._stream) == 0:
            return _OK_PAIR
        d = self._run(src, dst, angle, shift_x, shift_y)
        return d.success, d.info

    def Run(self, src: Surface, dst: Surface, angle: float, shift_x: float = 0.0,
            shift_y: float = 0.0) -> Tuple[bool, TaskExecInfo]:
        r = self.RunAsync(src, dst, angle, shift_x, shift_y)
        self._sync()
        return r


# ---- PySurfaceResizer --------------------------------------------------------------------
_RESIZE_FORMATS = (F.RGB, F.BGR, F.YUV420, F.YUV444, F.RGB_PLANAR, F.RGB_32F, F.RGB_32F_PLANAR, F.NV12,
                   # beyond the reference's list (TaskResizeSurface.cpp:293-309): same kernel
                   F.Y, F.P10, F.P12, F.YUV422, F.YUV420_10bit, F.YUV444_10bit)


class PySurfaceResizer(_SurfaceTask):
    """Resize a Surface to the size of the destination Surface.

    reference: src/python_vali/src/PySurfaceResizer.cpp:28-147; ResizeSurface
    (src/TC/src/TaskResizeSurface.cpp:293-328).  One launch resizes every plane (NV12 needs
    five NPP launches and two temporaries in the reference).  The default filter is the reference's:
    every nppiResize call site passes NPPI_INTER_LANCZOS (TaskResizeSurface.cpp:67,116,224,273), so
    `PySurfaceResizer(format, gpu_id[, stream])` is the 6x6 Lanczos-3 -- 3 lobes, normalised taps on the
    grid src = dst * scale, pinned against NPP output at a non-integer ratio at 45.6 dB
    (tests/test_oracle_reference_pins.py).  `interpolation=Interpolation.LINEAR` selects the bilinear
    filter BASELINE.json config 3 names, `Interpolation.CUBIC` the 4x4 Catmull-Rom bicubic (both
    extensions: the reference has no switch).
    RGB_PLANAR: the reference resizes the 3 stacked planes as ONE W x 3H image so rows
    bleed across channel seams (TaskResizeSurface.cpp:298, Surfaces.hpp:409); here each
    channel is resized on its own.
    """

    def __init__(self, format: PixelFormat, gpu_id: int, stream=None,
                 interpolation: Interpolation = Interpolation.LANCZOS):
        fmt = PixelFormat(format)
        if fmt not in _RESIZE_FORMATS:                       # TaskResizeSurface.cpp:307-308
            raise RuntimeError("pixel format not supported")
        super().__init__(gpu_id, stream)
        self._format = fmt
        self._interp = int(Interpolation(interpolation))

    @property
    def Interpolation(self) -> Interpolation:
        return Interpolation(self._interp)

    @property
    def Format(self) -> PixelFormat:
        return self._format

    def _run(self, src: Surface, dst: Surface) -> TaskExecDetails:
        if src is None or dst is None or src.IsEmpty or dst.IsEmpty:
            return _S_INVALID
        if dst.Format != src.Format or src.Format != self._format:   # :46-48, :93-95
            return _S_INVALID
        d = _status(shim.resize(src.desc(), dst.desc(), self._interp, self._stream))
        if d is _S_OK:
            self._memo_put((src.desc(), dst.desc()), shim.resize, (self._interp,))
        return d

    def RunAsync(self, src: Surface, dst: Surface) -> Tuple[bool, TaskExecInfo]:
        try:
            d1, d2 = src._desc, dst._desc
            m = self._memo.get((d1, d2))
        except (AttributeError, TypeError):
            m = None
        if m is not None and m[0](d1, d2, *m[1], self._stream) == 0:
            return _OK_PAIR
        d = self._run(src, dst)
        return d.success, d.info

    def Run(self, src: Surface, dst: Surface) -> Tuple[bool, TaskExecInfo]:
        r = self.RunAsync(src, dst)
        self._sync()
        return r

    def PrepareBatch(self, srcs: Sequence[Surface], dsts: Sequence[Surface]) -> "SurfaceBatch":
        return SurfaceBatch(self._gpu_id, self._stream, srcs, dsts)

    def RunBatchAsync(self, batch, dsts=None) -> Tuple[bool, TaskExecInfo]:
        batch = self._batch_of(batch, dsts)
        if batch.src_format != batch.dst_format or batch.src_format != self._format:
            return False, TaskExecInfo.INVALID_INPUT
        d = _status(shim.resize_batch(batch.d_src, batch.d_dst, batch.n, int(self._format),
                                      batch.src_size[0], batch.src_size[1],
                                      batch.dst_size[0], batch.dst_size[1], self._interp,
                                      self._stream))
        return d.success, d.info

    def RunBatch(self, batch, dsts=None) -> Tuple[bool, TaskExecInfo]:
        r = self.RunBatchAsync(batch, dsts)
        self._sync()
        return r
