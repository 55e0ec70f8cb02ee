"""ctypes front-end of the C oracle (oracle/vali_oracle.c) with numpy in/out.

TEST INFRASTRUCTURE ONLY -- may be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by vali_amd.  Every function operates on HOST arrays.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "libvali_oracle.so"

FMT = dict(Y=1, RGB=2, NV12=3, YUV420=4, RGB_PLANAR=5, BGR=6, YUV444=7, RGB_32F=8,
           RGB_32F_PLANAR=9, YUV422=10, P10=11, P12=12, YUV444_10bit=13, YUV420_10bit=14)

CSC_YUV, CSC_709CSC, CSC_709HDTV, CSC_YCBCR = 0, 1, 2, 3


class Surface(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("pitch", C.c_int32 * 3), ("width", C.c_int32),
                ("height", C.c_int32), ("format", C.c_int32)]


class Csc(C.Structure):
    _fields_ = [("y0", C.c_float), ("cy", C.c_float), ("crv", C.c_float), ("cgu", C.c_float),
                ("cgv", C.c_float), ("cbu", C.c_float), ("reserved", C.c_float * 2)]

    def astuple(self):
        return (self.y0, self.cy, self.crv, self.cgu, self.cgv, self.cbu)


def build(force: bool = False) -> Path:
    deps = list(HERE.glob("vali_oracle*.c")) + [HERE / "vali_oracle.h"]
    if force or not LIB.exists() or LIB.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["make", "-C", str(HERE), "-s", "libvali_oracle.so"], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB))
        _lib.vali_oracle_q_u8.restype = C.c_uint8
        _lib.vali_oracle_q_u8.argtypes = [C.c_float]
    return _lib


def csc(variant: int) -> Csc:
    k = Csc()
    rc = lib().vali_oracle_csc(variant, C.byref(k))
    if rc:
        raise ValueError(f"unknown csc variant {variant}")
    return k


def csc_from_tuple(t) -> Csc:
    k = Csc()
    k.y0, k.cy, k.crv, k.cgu, k.cgv, k.cbu = t
    return k


def q_u8(values) -> np.ndarray:
    v = np.asarray(values, dtype=np.float32).ravel()
    f = lib().vali_oracle_q_u8
    return np.array([f(float(x)) for x in v], dtype=np.uint8)


def _ptr(a: np.ndarray, byte_offset: int = 0) -> int:
    return a.ctypes.data + byte_offset


def surf_nv12(a: np.ndarray, width: int, height: int) -> Surface:
    """a: (height*3/2, pitch) uint8 array holding Y rows then interleaved UV rows."""
    assert a.dtype == np.uint8 and a.ndim == 2 and a.flags.c_contiguous
    s = Surface()
    pitch = a.strides[0]
    s.plane[0] = _ptr(a)
    s.plane[1] = _ptr(a, height * pitch)
    s.pitch[0] = s.pitch[1] = pitch
    s.width, s.height, s.format = width, height, FMT["NV12"]
    return s


def surf_packed3(a: np.ndarray, width: int, height: int, fmt: str) -> Surface:
    assert a.ndim == 2 and a.flags.c_contiguous
    s = Surface()
    s.plane[0] = _ptr(a)
    s.pitch[0] = a.strides[0]
    s.width, s.height, s.format = width, height, FMT[fmt]
    return s


def surf_planar3(a: np.ndarray, width: int, height: int, fmt: str) -> Surface:
    """a: (3*height, pitch_elems) stacked planes in one allocation."""
    assert a.ndim == 2 and a.flags.c_contiguous
    s = Surface()
    pitch = a.strides[0]
    for c in range(3):
        s.plane[c] = _ptr(a, c * height * pitch)
        s.pitch[c] = pitch
    s.width, s.height, s.format = width, height, FMT[fmt]
    return s


def nv12_to_rgb(nv12: np.ndarray, width: int, height: int, k: Csc, dst_fmt: str = "RGB"
                ) -> np.ndarray:
    """nv12: (height*3/2, >=width) uint8.  Returns (H, 3W) packed or (3H, W) planar uint8."""
    src = surf_nv12(nv12, width, height)
    if dst_fmt == "RGB_PLANAR":
        out = np.zeros((3 * height, width), np.uint8)
        dst = surf_planar3(out, width, height, dst_fmt)
    else:
        out = np.zeros((height, 3 * width), np.uint8)
        dst = surf_packed3(out, width, height, dst_fmt)
    rc = lib().vali_oracle_nv12_to_rgb(C.byref(src), C.byref(dst), C.byref(k))
    if rc:
        raise RuntimeError(f"vali_oracle_nv12_to_rgb -> {rc}")
    return out


def nv12_to_rgb_mt(frames, width, height, k: Csc, threads: int, outs=None, simd: bool = False):
    """frames: list of NV12 arrays; converts all with `threads` OpenMP threads (baseline).
    simd=True: the AVX2 form (vali_oracle_simd.c), bit-identical output."""
    n = len(frames)
    outs = outs if outs is not None else [np.zeros((height, 3 * width), np.uint8) for _ in range(n)]
    S = (Surface * n)(*[surf_nv12(f, width, height) for f in frames])
    D = (Surface * n)(*[surf_packed3(o, width, height, "RGB") for o in outs])
    fn = lib().vali_oracle_nv12_to_rgb_simd_mt if simd else lib().vali_oracle_nv12_to_rgb_mt
    rc = fn(S, D, n, C.byref(k), int(threads))
    if rc:
        raise RuntimeError(f"vali_oracle_nv12_to_rgb_mt -> {rc}")
    return outs


def nv12_to_rgb_bench(frame: np.ndarray, width: int, height: int, k: Csc, threads: int, seconds: float, simd: bool = True,
                      cpus=None):
    """(frames converted, seconds): `threads` OpenMP threads, each on its own first-touched copy of `frame` and pinned to
    cpus[i] (NUMA-fair CPU baseline of bench.py; vali_oracle_simd.c)."""
    fn = lib().vali_oracle_nv12_to_rgb_bench
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double),
                   C.POINTER(C.c_int), C.c_int]
    a = np.ascontiguousarray(frame, np.uint8)
    dt = C.c_double(0.0)
    ids = (C.c_int * len(cpus))(*cpus) if cpus else None
    n = fn(a.ctypes.data, int(width), int(height), C.cast(C.byref(k), C.c_void_p), int(threads), float(seconds), int(simd),
           C.byref(dt), ids, len(cpus) if cpus else 0)
    if n < 0:
        raise RuntimeError("vali_oracle_nv12_to_rgb_bench: bad arguments")
    return int(n), float(dt.value)


def surf_semiplanar(a: np.ndarray, width: int, height: int, fmt: str) -> Surface:
    """NV12 (uint8) or P10/P12 (uint16) array of shape (height*3/2, >=width)."""
    assert a.ndim == 2 and a.flags.c_contiguous
    s = Surface()
    pitch = a.strides[0]
    s.plane[0] = _ptr(a)
    s.plane[1] = _ptr(a, height * pitch)
    s.pitch[0] = s.pitch[1] = pitch
    s.width, s.height, s.format = width, height, FMT[fmt]
    return s


def surf_planes3(planes, width: int, height: int, fmt: str) -> Surface:
    """three separate 2-D arrays (YUV444 / YUV420 ...)."""
    s = Surface()
    for c, a in enumerate(planes):
        assert a.ndim == 2 and a.flags.c_contiguous
        s.plane[c] = _ptr(a)
        s.pitch[c] = a.strides[0]
    s.width, s.height, s.format = width, height, FMT[fmt]
    return s


_UD_DTYPE = {"YUV444": np.uint8, "YUV444_10bit": np.uint16, "RGB": np.uint8, "RGB_PLANAR": np.uint8,
             "RGB_32F": np.float32, "RGB_32F_PLANAR": np.float32}


def ud_nv12(src: np.ndarray, sw: int, sh: int, src_fmt: str, dw: int, dh: int, dst_fmt: str):
    """Returns the dst image in the reference's host layout: YUV444 -> (3, dh, dw) stacked
    planes, packed RGB -> (dh, 3*dw), planar RGB -> (3*dh, dw)."""
    s = surf_semiplanar(src, sw, sh, src_fmt)
    dt = _UD_DTYPE[dst_fmt]
    if dst_fmt in ("YUV444", "YUV444_10bit"):
        out = np.zeros((3, dh, dw), dt)
        d = surf_planes3([out[0], out[1], out[2]], dw, dh, dst_fmt)
    elif dst_fmt in ("RGB", "RGB_32F"):
        out = np.zeros((dh, 3 * dw), dt)
        d = surf_packed3(out, dw, dh, dst_fmt)
    else:
        out = np.zeros((3 * dh, dw), dt)
        d = surf_planar3(out, dw, dh, dst_fmt)
    rc = lib().vali_oracle_ud_nv12(C.byref(s), C.byref(d))
    if rc:
        raise RuntimeError(f"vali_oracle_ud_nv12 -> {rc}")
    return out


def rotate_plane(src: np.ndarray, channels: int, dst_w: int, dst_h: int, angle: float,
                 shift_x: float = 0.0, shift_y: float = 0.0, fill=0) -> np.ndarray:
    """src: (H, W*channels) array of uint8/uint16/float32.  Untouched dst pixels keep `fill`."""
    assert src.ndim == 2 and src.flags.c_contiguous
    elem = src.dtype.itemsize
    sh, sw = src.shape[0], src.shape[1] // channels
    out = np.full((dst_h, dst_w * channels), fill, src.dtype)
    rc = lib().vali_oracle_rotate_plane(C.c_void_p(src.ctypes.data), src.strides[0], sw, sh,
                                        C.c_void_p(out.ctypes.data), out.strides[0], dst_w, dst_h,
                                        elem, channels, C.c_double(angle), C.c_double(shift_x),
                                        C.c_double(shift_y))
    if rc:
        raise RuntimeError(f"vali_oracle_rotate_plane -> {rc}")
    return out


def canonical_shifts(angle: float, src_w: int, src_h: int):
    """Shift normalisation of PySurfaceRotator::Run (PySurfaceRotator.cpp:47-73)."""
    n = (int(round(angle)) + 360) % 360
    return {0: (0.0, 0.0, 0.0), 90: (90.0, 0.0, src_w - 1.0), 180: (180.0, src_w - 1.0, src_h - 1.0),
            270: (270.0, src_h - 1.0, 0.0)}[n]


def resize_plane(src: np.ndarray, channels: int, dst_w: int, dst_h: int, interp: str = "linear") -> np.ndarray:
    """src: (H, W*channels) uint8/uint16/float32 -> (dst_h, dst_w*channels); linear | cubic | lanczos."""
    assert src.ndim == 2 and src.flags.c_contiguous
    sh, sw = src.shape[0], src.shape[1] // channels
    out = np.zeros((dst_h, dst_w * channels), src.dtype)
    fn = {"linear": lib().vali_oracle_resize_plane, "lanczos": lib().vali_oracle_resize_plane_lanczos,
          "cubic": lib().vali_oracle_resize_plane_cubic}[interp]
    rc = fn(C.c_void_p(src.ctypes.data), src.strides[0], sw, sh,
            C.c_void_p(out.ctypes.data), out.strides[0], dst_w, dst_h,
            src.dtype.itemsize, channels)
    if rc:
        raise RuntimeError(f"vali_oracle_resize_plane[{interp}] -> {rc}")
    return out


def lanczos3_weights(a: float) -> np.ndarray:
    w = (C.c_float * 6)()
    f = lib().vali_oracle_lanczos3_weights
    f.restype = None
    f.argtypes = [C.c_float, C.POINTER(C.c_float)]
    f(a, w)
    return np.array(list(w), np.float32)


def cubic_weights(a: float) -> np.ndarray:
    w = (C.c_float * 4)()
    f = lib().vali_oracle_cubic_weights
    f.restype = None
    f.argtypes = [C.c_float, C.POINTER(C.c_float)]
    f(a, w)
    return np.array(list(w), np.float32)


# plane lists in HOST (tightly packed) layout: (rows_fn, cols_fn, channels, subsample_x, subsample_y)
def host_planes(fmt: str, w: int, h: int):
    """[(plane_w_pixels, plane_h, channels)] in upload order for a format (vali_amd.surface.FORMATS)."""
    if fmt in ("NV12", "P10", "P12"):
        return [(w, h, 1), (w // 2, h // 2, 2)]
    if fmt in ("YUV420", "YUV420_10bit"):
        return [(w, h, 1), (w // 2, h // 2, 1), (w // 2, h // 2, 1)]
    if fmt == "YUV422":
        return [(w, h, 1), (w // 2, h, 1), (w // 2, h, 1)]
    if fmt in ("YUV444", "YUV444_10bit", "RGB_PLANAR", "RGB_32F_PLANAR"):
        return [(w, h, 1)] * 3
    if fmt in ("RGB", "BGR", "RGB_32F"):
        return [(w, h, 3)]
    if fmt == "Y":
        return [(w, h, 1)]
    raise ValueError(fmt)


def resize_surface(host: np.ndarray, fmt: str, sw: int, sh: int, dw: int, dh: int,
                   interp: str = "linear") -> np.ndarray:
    """Resize a whole surface given as its flat host image (element dtype = host.dtype)."""
    out, off = [], 0
    for (pw, ph, ch), (qw, qh, _) in zip(host_planes(fmt, sw, sh), host_planes(fmt, dw, dh)):
        n = pw * ph * ch
        plane = np.ascontiguousarray(host[off: off + n].reshape(ph, pw * ch))
        out.append(resize_plane(plane, ch, qw, qh, interp).reshape(-1))
        off += n
    return np.concatenate(out)


class CvtParams(C.Structure):
    _fields_ = [("yuv2rgb", Csc), ("rgb2yuv", (C.c_float * 4) * 3)]


def cvt_params(csc_variant=None, rgb2yuv_variant=None, csc_tuple=None, rgb2yuv_rows=None) -> CvtParams:
    p = CvtParams()
    if csc_tuple is not None:
        p.yuv2rgb = csc_from_tuple(csc_tuple)
    elif csc_variant is not None:
        p.yuv2rgb = csc(csc_variant)
    if rgb2yuv_rows is not None:
        for i in range(3):
            for j in range(4):
                p.rgb2yuv[i][j] = rgb2yuv_rows[i][j]
    elif rgb2yuv_variant is not None:
        rc = lib().vali_oracle_rgb2yuv(rgb2yuv_variant, p.rgb2yuv)
        if rc:
            raise ValueError("bad rgb2yuv variant")
    return p


def rgb2yuv_rows(variant: int):
    p = cvt_params(rgb2yuv_variant=variant)
    return tuple(tuple(float(p.rgb2yuv[i][j]) for j in range(4)) for i in range(3))


_ELEM = {"P10": np.uint16, "P12": np.uint16, "RGB_32F": np.float32, "RGB_32F_PLANAR": np.float32,
         "YUV444_10bit": np.uint16, "YUV420_10bit": np.uint16}


def surface_from_host(host: np.ndarray, fmt: str, w: int, h: int):
    """Wrap a flat tightly-packed host image (upload layout) as an oracle Surface.
    Returns (Surface, keepalive)."""
    dt = _ELEM.get(fmt, np.uint8)
    flat = np.ascontiguousarray(host).view(dt).reshape(-1)
    s = Surface()
    s.width, s.height, s.format = w, h, FMT[fmt]
    e = np.dtype(dt).itemsize
    base = flat.ctypes.data
    if fmt in ("NV12", "P10", "P12"):
        s.plane[0], s.plane[1] = base, base + w * h * e
        s.pitch[0] = s.pitch[1] = w * e
    elif fmt in ("RGB_PLANAR", "RGB_32F_PLANAR"):
        for c in range(3):
            s.plane[c] = base + c * w * h * e
            s.pitch[c] = w * e
    elif fmt in ("RGB", "BGR", "RGB_32F"):
        s.plane[0] = base
        s.pitch[0] = 3 * w * e
    else:
        off = 0
        for c, (pw, ph, ch) in enumerate(host_planes(fmt, w, h)):
            s.plane[c] = base + off
            s.pitch[c] = pw * ch * e
            off += pw * ph * ch * e
    return s, flat


def host_size(fmt: str, w: int, h: int) -> int:
    e = np.dtype(_ELEM.get(fmt, np.uint8)).itemsize
    return sum(pw * ph * ch for pw, ph, ch in host_planes(fmt, w, h)) * e


def convert(host: np.ndarray, src_fmt: str, dst_fmt: str, w: int, h: int, params: CvtParams) -> np.ndarray:
    """Generic converter on flat host images; returns the flat dst image (uint8 view)."""
    s, keep = surface_from_host(host, src_fmt, w, h)
    out = np.zeros(host_size(dst_fmt, w, h), np.uint8)
    d, keep2 = surface_from_host(out, dst_fmt, w, h)
    rc = lib().vali_oracle_convert(C.byref(s), C.byref(d), C.byref(params))
    if rc:
        raise RuntimeError(f"vali_oracle_convert {src_fmt}->{dst_fmt} -> {rc}")
    return out
