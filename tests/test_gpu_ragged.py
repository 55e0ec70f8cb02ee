"""Ragged widths and foreign alignment on the converters: the vector body + byte-granular tail.

NPP accepts any ROI, base pointer and step (reference call sites: src/TC/src/TaskConvertSurface.cpp:120-136
hand `Surface::PixelPtr`, `Pitch` and an NppiSize straight through), so a drop-in must convert 854x480,
1366x768, 1918x1078 and tensors with odd strides at full speed AND without touching a byte outside the
rows it was given.  Every case is bit-exact against the C oracle; the misaligned cases run through the raw
C ABI on planes placed at odd offsets / odd pitches inside one sentinel-filled device buffer, which is
downloaded whole afterwards: everything outside the destination rows must still hold the sentinel.
4:2:0 surfaces with an odd width or height are refused (the chroma planes hold (W/2) x (H/2) samples).
"""
import numpy as np
import pytest

from conftest import make_nv12

pytestmark = pytest.mark.gpu
SENTINEL = 0xA5


def _elem(fmt):
    return {"P10": 2, "P12": 2, "RGB_32F": 4, "RGB_32F_PLANAR": 4}.get(fmt, 1)


class Arena:
    """One device buffer filled with a sentinel; planes are carved out at chosen offsets / pitches."""

    def __init__(self, vali, gpu, size):
        from vali_amd._native import shim

        self.shim, self.gpu, self.size = shim, gpu, size
        self.ptr = shim.mem_alloc(gpu, size)
        shim.memset2d_async(gpu, self.ptr, size, SENTINEL, size, 1, 0)
        shim.stream_sync(gpu, 0)
        self.cursor = 0
        self.regions = []          # (offset, pitch, row_bytes, rows) of every carved plane

    def carve(self, row_bytes, rows, misalign, extra_pitch):
        off = ((self.cursor + 255) // 256) * 256 + misalign
        pitch = row_bytes + extra_pitch
        self.cursor = off + pitch * rows + 64
        assert self.cursor <= self.size
        self.regions.append((off, pitch, row_bytes, rows))
        return off, pitch

    def upload(self, off, pitch, data2d):
        a = np.ascontiguousarray(data2d)
        host_ptr, _, _ = self.shim.buffer_info(a, False)
        self.shim.memcpy2d_async(self.gpu, self.ptr + off, pitch, host_ptr, a.shape[1], a.shape[1], a.shape[0], 0, 0)
        self.shim.stream_sync(self.gpu, 0)

    def download_all(self):
        out = np.zeros(self.size, np.uint8)
        host_ptr, _, _ = self.shim.buffer_info(out, True)
        self.shim.memcpy2d_async(self.gpu, host_ptr, self.size, self.ptr, self.size, self.size, 1, 1, 0)
        self.shim.stream_sync(self.gpu, 0)
        return out

    def free(self):
        self.shim.mem_free(self.gpu, self.ptr)


def place(arena, oracle, fmt, w, h, host, misalign, extra_pitch):
    """Carve the planes of a tightly packed host image (upload layout) into the arena.
    Returns (SurfaceDesc, [(offset, pitch, row_bytes, rows)] in host-plane order)."""
    from vali_amd._native import shim

    e = _elem(fmt)
    planes, ptrs, pitches, pos = [], [], [], 0
    flat = None if host is None else np.ascontiguousarray(host).view(np.uint8).reshape(-1)
    for pw, ph, ch in oracle.host_planes(fmt, w, h):
        rb = pw * ch * e
        off, pitch = arena.carve(rb, ph, misalign * e, extra_pitch * e)
        if host is not None:
            arena.upload(off, pitch, flat[pos:pos + rb * ph].reshape(ph, rb))
        pos += rb * ph
        planes.append((off, pitch, rb, ph))
        ptrs.append(arena.ptr + off)
        pitches.append(pitch)
    if fmt in ("RGB_PLANAR", "RGB_32F_PLANAR"):     # one stacked allocation in the product; here 3 planes, same pitch
        assert len(set(pitches)) == 1
    return shim.SurfaceDesc(ptrs, pitches, w, h, int(oracle.FMT[fmt])), planes


def gather(buf, planes):
    return np.concatenate([buf[off + r * pitch: off + r * pitch + rb] for off, pitch, rb, rows in planes
                           for r in range(rows)])


def check_sentinel(buf, arena, written):
    mask = np.ones(arena.size, bool)
    for off, pitch, rb, rows in written:
        for r in range(rows):
            mask[off + r * pitch: off + r * pitch + rb] = False
    for off, pitch, rb, rows in arena.regions:          # source planes were uploaded: not sentinel either
        if (off, pitch, rb, rows) not in written:
            for r in range(rows):
                mask[off + r * pitch: off + r * pitch + rb] = False
    assert np.all(buf[mask] == SENTINEL), "bytes outside the destination rows were written"


def host_image(fmt, w, h, seed, oracle):
    rng = np.random.default_rng(seed)
    n = oracle.host_size(fmt, w, h)
    if fmt in ("P10", "P12"):
        return (rng.integers(0, 1024, n // 2, dtype=np.uint16) << 6).astype(np.uint16).view(np.uint8)
    if fmt == "RGB_32F":
        return rng.random(n // 4, dtype=np.float32).view(np.uint8)
    return rng.integers(0, 256, n, dtype=np.uint8)


RAGGED = [(854, 480), (1366, 768), (1918, 1078), (1026, 4), (18, 2), (4098, 6), (30, 2)]


@pytest.mark.parametrize("size", RAGGED)
@pytest.mark.parametrize("dst", ["RGB", "BGR", "RGB_PLANAR"])
def test_nv12_rgb_ragged_widths(vali, gpu, oracle, size, dst):
    w, h = size
    nv12 = make_nv12(w, h, 11)
    s = vali.Surface.Make(vali.NV12, w, h, gpu)
    d = vali.Surface.Make(vali.PixelFormat[dst], w, h, gpu)
    assert vali.PyFrameUploader(gpu).Run(nv12.reshape(-1), s)[0]
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    assert vali.PySurfaceConverter(gpu).Run(s, d, cc) == (True, vali.TaskExecInfo.SUCCESS)
    got = np.zeros(d.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(d, got)[0]
    assert np.array_equal(got, oracle.nv12_to_rgb(nv12, w, h, oracle.csc(1), dst).reshape(-1))


def _pairs():
    import vali_amd as vali

    return [(s.name, d.name) for s, d in vali.PySurfaceConverter.Conversions()]


def _params(vali, oracle, src, dst):
    """(shim params, oracle params) of the pair's DEFAULT colour context."""
    from vali_amd import tasks

    yuv = ("NV12", "YUV420", "YUV444")
    rgb = ("RGB", "BGR", "RGB_PLANAR")
    if src == "NV12" and dst in rgb:
        return tasks._params(csc=tasks.CSC_NPP_709HDTV), oracle.cvt_params(csc_variant=2)
    if src in yuv and dst in rgb:
        return tasks._params(csc=tasks.CSC_NPP_YUV), oracle.cvt_params(csc_variant=0)
    if src in rgb and dst in ("YUV444", "YUV420", "Y"):
        return tasks._params(rgb2yuv=tasks.RGB2YUV_NPP_YUV), oracle.cvt_params(rgb2yuv_variant=0)
    return tasks._params(), oracle.cvt_params()


@pytest.mark.parametrize("pair", _pairs(), ids=lambda p: f"{p[0]}-{p[1]}")
@pytest.mark.parametrize("geom", [(854, 480, 1, 3), (70, 6, 5, 1), (1366, 10, 3, 7), (1040, 4, 9, 0)],
                         ids=lambda g: f"{g[0]}x{g[1]}+{g[2]}p{g[3]}")
def test_every_pair_on_misaligned_planes(vali, gpu, oracle, pair, geom):
    """C ABI, planes at base offsets `misalign` elements off a 256-byte boundary and pitches `extra` elements
    longer than the row: nothing is 16-byte aligned, widths are not multiples of 16."""
    from vali_amd._native import shim

    src, dst = pair
    w, h, misalign, extra = geom
    host = host_image(src, w, h, 21, oracle)
    arena = Arena(vali, gpu, 2 * (oracle.host_size(src, w, h) + oracle.host_size(dst, w, h)) + (1 << 16) + 64 * h * 8)
    try:
        sd, _ = place(arena, oracle, src, w, h, host, misalign, extra)
        dd, dplanes = place(arena, oracle, dst, w, h, None, misalign + 2, extra + 2 if extra else 0)
        sp, op = _params(vali, oracle, src, dst)
        if src == "NV12" and dst in ("RGB", "BGR", "RGB_PLANAR"):
            from vali_amd import tasks
            rc = shim.nv12_to_rgb(sd, dd, tasks._csc(tasks.CSC_NPP_709HDTV), 0)
        else:
            rc = shim.convert(sd, dd, sp, 0)
        assert rc == 0, shim.last_error()
        shim.stream_sync(gpu, 0)
        buf = arena.download_all()
        want = oracle.convert(host, src, dst, w, h, op)
        assert np.array_equal(gather(buf, dplanes), want)
        check_sentinel(buf, arena, dplanes)
    finally:
        arena.free()


NO_420 = [("RGB", "RGB_PLANAR"), ("RGB_PLANAR", "RGB"), ("RGB", "BGR"), ("RGB", "YUV444"), ("YUV444", "RGB"),
          ("YUV444", "BGR"), ("BGR", "YUV444"), ("RGB_PLANAR", "YUV444"), ("RGB", "Y"), ("Y", "YUV444"),
          ("RGB", "RGB_32F"), ("RGB_32F", "RGB_32F_PLANAR")]


@pytest.mark.parametrize("pair", NO_420, ids=lambda p: f"{p[0]}-{p[1]}")
@pytest.mark.parametrize("size", [(51, 35), (1919, 3), (1, 1), (17, 1), (1025, 5)])
def test_pairs_without_subsampling_take_odd_sizes(vali, gpu, oracle, pair, size):
    from test_gpu_convert import run_pair

    run_pair(vali, gpu, oracle, pair[0], pair[1], *size, seed=4)


@pytest.mark.parametrize("pair", [("NV12", "RGB"), ("NV12", "RGB_PLANAR"), ("NV12", "YUV420"), ("YUV420", "NV12"),
                                  ("RGB", "YUV420"), ("YUV420", "RGB"), ("P10", "NV12"), ("NV12", "Y")])
@pytest.mark.parametrize("size", [(64, 5), (63, 4), (65, 33)])
def test_subsampled_formats_refuse_odd_sizes(vali, gpu, pair, size):
    """RGB 64x5 -> YUV420 would store a third chroma row into a 2-row plane (the planes hold (W/2) x (H/2)
    samples, Surfaces.cpp:231-246): refused with INVALID_INPUT, nothing is launched."""
    w, h = size
    s = vali.Surface.Make(vali.PixelFormat[pair[0]], w, h, gpu)
    d = vali.Surface.Make(vali.PixelFormat[pair[1]], w, h, gpu)
    if pair == ("P10", "NV12") and w % 2 == 0:
        # a semi-planar Surface derives its height from its single plane (rows * 2 // 3, Surfaces.cpp:104-113):
        # Make(P10, 64, 5) IS a 64x4 surface on both sides -- nothing odd reaches the converter
        assert s.Height == h - 1 and d.Height == h - 1
        return
    assert vali.PySurfaceConverter(gpu).Run(s, d) == (False, vali.TaskExecInfo.INVALID_INPUT)
    assert vali.PySurfaceConverter(gpu).RunBatch([s], [d]) == (False, vali.TaskExecInfo.INVALID_INPUT)


def test_c_abi_refuses_odd_sizes_for_subsampled_formats(vali, gpu):
    from vali_amd._native import shim
    from vali_amd import tasks

    buf = shim.mem_alloc(gpu, 1 << 20)
    try:
        for sf, df in ((11, 3), (3, 4), (2, 4), (3, 1)):          # P10->NV12, NV12->YUV420, RGB->YUV420, NV12->Y
            for w, h in ((64, 5), (63, 4)):
                a = shim.SurfaceDesc([buf, buf + 65536, buf + 131072], [256, 256, 256], w, h, sf)
                b = shim.SurfaceDesc([buf + 262144, buf + 327680, buf + 393216], [256, 256, 256], w, h, df)
                assert shim.convert(a, b, tasks._params(rgb2yuv=tasks.RGB2YUV_NPP_YUV), 0) == shim.ERR_INVALID_ARG
                assert "even width and height" in shim.last_error()
        a = shim.SurfaceDesc([buf, buf + 65536], [256, 256], 64, 5, 3)
        b = shim.SurfaceDesc([buf + 262144], [256], 64, 5, 2)
        assert shim.nv12_to_rgb(a, b, tasks._csc(tasks.CSC_NPP_709CSC), 0) == shim.ERR_INVALID_ARG
    finally:
        shim.mem_free(gpu, buf)


def test_odd_stride_torch_tensors_both_sides(vali, gpu, oracle):
    """NV12 source and RGB destination are both torch tensors with odd row strides (from_dlpack)."""
    import torch

    w, h = 1366, 768
    nv12 = make_nv12(w, h, 2)
    sbuf = torch.zeros((h * 3 // 2, w + 3), dtype=torch.uint8, device="cuda")
    sbuf[:, :w].copy_(torch.from_numpy(nv12))
    dbuf = torch.full((h, 3 * w + 5), SENTINEL, dtype=torch.uint8, device="cuda")
    src = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(sbuf[:, :w]), vali.NV12)
    dst = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(dbuf[:, :3 * w]), vali.RGB)
    assert src.Pitch == w + 3 and dst.Pitch == 3 * w + 5 and dst.Width == w
    torch.cuda.synchronize()
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    assert vali.PySurfaceConverter(gpu).Run(src, dst, cc)[0]
    out = dbuf.cpu().numpy()
    assert np.array_equal(out[:, :3 * w], oracle.nv12_to_rgb(nv12, w, h, oracle.csc(1), "RGB"))
    assert np.all(out[:, 3 * w:] == SENTINEL)


@pytest.mark.parametrize("pair", [("NV12", "RGB"), ("NV12", "RGB_PLANAR"), ("RGB", "YUV420"), ("RGB", "RGB_PLANAR"), ("YUV420", "BGR")])
def test_ragged_batch_equals_oracle(vali, gpu, oracle, pair):
    """the batched launches (device descriptor arrays) take the same ragged path, decided per frame in the kernel"""
    src, dst = pair
    w, h, n = 854, 480, 5
    cvt = vali.PySurfaceConverter(gpu)
    srcs = [vali.Surface.Make(vali.PixelFormat[src], w, h, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(vali.PixelFormat[dst], w, h, gpu) for _ in range(n)]
    hosts = [host_image(src, w, h, 60 + i, oracle) for i in range(n)]
    for hst, s in zip(hosts, srcs):
        assert vali.PyFrameUploader(gpu).Run(hst, s)[0]
    assert cvt.RunBatch(srcs, dsts) == (True, vali.TaskExecInfo.SUCCESS)
    _, op = _params(vali, oracle, src, dst)
    for hst, d in zip(hosts, dsts):
        got = np.zeros(d.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(d, got)[0]
        assert np.array_equal(got, oracle.convert(hst, src, dst, w, h, op))
