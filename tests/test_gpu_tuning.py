"""The tuning table (include/vali_hip.h: vali_tuning_key) cannot change results: every operator is run
in-process under every value of the switches that select its kernel form and must reproduce the default
output byte for byte.  Also: tracing on (a roctx range per entry point) leaves results alone."""
import numpy as np
import pytest

from conftest import make_nv12

pytestmark = pytest.mark.gpu


def upload(vali, gpu, fmt, w, h, host):
    s = vali.Surface.Make(fmt, w, h, gpu)
    assert vali.PyFrameUploader(gpu).Run(np.ascontiguousarray(host).reshape(-1), s)[0]
    return s


def download(vali, gpu, surf):
    out = np.zeros(surf.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(surf, out)[0]
    return out


def nv12_rgb(vali, gpu, w, h, dst):
    src = upload(vali, gpu, vali.NV12, w, h, make_nv12(w, h, 1))
    d = vali.Surface.Make(dst, w, h, gpu)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    assert vali.PySurfaceConverter(gpu).Run(src, d, cc)[0]
    return download(vali, gpu, d)


def resize(vali, gpu, sw, sh, dw, dh, interp):
    src = upload(vali, gpu, vali.NV12, sw, sh, make_nv12(sw, sh, 2))
    d = vali.Surface.Make(vali.NV12, dw, dh, gpu)
    assert vali.PySurfaceResizer(vali.NV12, gpu, interpolation=interp).Run(src, d)[0]
    return download(vali, gpu, d)


def ud(vali, gpu, sw, sh, dw, dh, dst):
    src = upload(vali, gpu, vali.NV12, sw, sh, make_nv12(sw, sh, 3))
    d = vali.Surface.Make(dst, dw, dh, gpu)
    assert vali.PySurfaceUD(gpu).Run(src, d)[0]
    return download(vali, gpu, d)


def preproc(vali, gpu, sw, sh, dw, dh):
    src = upload(vali, gpu, vali.NV12, sw, sh, make_nv12(sw, sh, 5))
    d = vali.Surface.Make(vali.RGB_32F_PLANAR, dw, dh, gpu)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    pp = vali.PySurfacePreprocessor(gpu, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), div=1.0)
    assert pp.Run(src, d, cc)[0]
    return download(vali, gpu, d)


def rotate(vali, gpu, w, h, angle):
    rng = np.random.default_rng(4)
    src = upload(vali, gpu, vali.RGB, w, h, rng.integers(0, 256, w * h * 3, dtype=np.uint8))
    q = int(angle) % 180 != 0
    d = vali.Surface.Make(vali.RGB, h if q else w, w if q else h, gpu)
    assert vali.PyFrameUploader(gpu).Run(np.full(d.HostSize, 9, np.uint8), d)[0]   # pixels that sample outside the source stay untouched
    sx, sy = (0.0, 0.0) if angle % 90.0 == 0.0 else (w * 0.3, h * 0.6)
    assert vali.PySurfaceRotator(gpu).Run(src, d, angle, sx, sy)[0]
    return download(vali, gpu, d)


CASES = [
    ("NV12_ROWPAIRS", (1, 2, 4), lambda v, g: nv12_rgb(v, g, 640, 360, v.RGB)),
    ("NV12_ROWPAIRS", (1, 2), lambda v, g: nv12_rgb(v, g, 1920, 1080, v.RGB_PLANAR)),
    ("WAVES_PER_CU", (4, 8, 24, 32), lambda v, g: nv12_rgb(v, g, 1920, 1080, v.RGB)),
    ("NV12_DIRECT_STORE", (1,), lambda v, g: nv12_rgb(v, g, 1920, 1080, v.BGR)),
    ("NV12_DIRECT_STORE", (1,), lambda v, g: nv12_rgb(v, g, 854, 480, v.RGB)),
    ("RESIZE_POINT", (0, 2), lambda v, g: resize(v, g, 1920, 1080, 640, 360, v.Interpolation.LINEAR)),
    ("RESIZE_POINT", (0, 2), lambda v, g: resize(v, g, 1920, 1080, 960, 540, v.Interpolation.LANCZOS)),
    ("RESIZE_POINT", (0, 2), lambda v, g: resize(v, g, 1272, 720, 318, 90, v.Interpolation.CUBIC)),
    ("RESIZE_FORCE_GATHER", (1,), lambda v, g: resize(v, g, 1280, 720, 854, 480, v.Interpolation.LINEAR)),
    ("RESIZE_FORCE_GATHER", (1,), lambda v, g: resize(v, g, 1280, 720, 854, 480, v.Interpolation.LANCZOS)),
    ("RESIZE_FORCE_GATHER", (1,), lambda v, g: resize(v, g, 640, 360, 1280, 720, v.Interpolation.CUBIC)),
    ("RESIZE_NO_SEPARABLE", (1, 2, 3), lambda v, g: resize(v, g, 1280, 720, 854, 480, v.Interpolation.LANCZOS)),
    ("RESIZE_NO_SEPARABLE", (1, 2, 3), lambda v, g: resize(v, g, 640, 360, 1280, 720, v.Interpolation.CUBIC)),
    ("RESIZE_NO_SEPARABLE", (4,), lambda v, g: resize(v, g, 640, 360, 800, 450, v.Interpolation.LANCZOS)),      # growing planes on 64-row waves
    ("RESIZE_NO_SEPARABLE", (11, 13, 22), lambda v, g: resize(v, g, 960, 540, 640, 360, v.Interpolation.LANCZOS)),  # 3:2 both ways: 1 / 3 / 12 row pairs per wave
    ("RESIZE_NO_SEPARABLE", (12, 31), lambda v, g: resize(v, g, 1280, 720, 644, 364, v.Interpolation.LANCZOS)),   # general form: 2 / 21 rows per slot
    ("RESIZE_NO_SEPARABLE", (11, 16), lambda v, g: resize(v, g, 1280, 720, 640, 364, v.Interpolation.LANCZOS)),   # 2:1 along x: 1 / 6 rows per slot
    # general-ratio shrinking planes: 1 both passes in every wave (round 4's form), 2 specialised waves without tap tables
    ("RESIZE_COLS", (1, 2), lambda v, g: resize(v, g, 1280, 720, 644, 364, v.Interpolation.LANCZOS)),
    ("RESIZE_COLS", (1, 2), lambda v, g: resize(v, g, 1280, 720, 854, 480, v.Interpolation.CUBIC)),      # 3 slots of 4 taps
    ("RESIZE_COLS", (1, 2), lambda v, g: resize(v, g, 960, 540, 800, 450, v.Interpolation.LANCZOS)),     # 6 slots
    ("RESIZE_ROWS", (0, 2, 3), lambda v, g: resize(v, g, 640, 360, 960, 540, v.Interpolation.LANCZOS)),
    ("RESIZE_ROWS", (0, 3), lambda v, g: resize(v, g, 640, 360, 800, 450, v.Interpolation.LANCZOS)),   # 3: the LDS-staged rows form instead of the register form
    ("RESIZE_ROWS", (0,), lambda v, g: resize(v, g, 640, 360, 1000, 700, v.Interpolation.CUBIC)),
    ("UD_DOWN2", (0, 2), lambda v, g: ud(v, g, 1280, 720, 640, 360, v.RGB)),
    ("UD_DOWN2", (0, 2), lambda v, g: ud(v, g, 1276, 720, 638, 360, v.RGB)),
    ("UD_DOWN2", (0, 2), lambda v, g: ud(v, g, 1280, 720, 1280, 720, v.RGB_PLANAR)),   # 2: k_ud_lean with the general rows
    ("UD_DOWN2", (0, 2), lambda v, g: ud(v, g, 1280, 720, 1280, 360, v.RGB)),
    ("UD_OCC5", (1,), lambda v, g: ud(v, g, 1280, 720, 854, 480, v.RGB)),     # (a no-op since round 2)
    ("UD_OCC5", (1,), lambda v, g: ud(v, g, 140, 108, 4, 57, v.RGB)),
    ("UD_FORCE_GATHER", (1,), lambda v, g: ud(v, g, 1280, 720, 854, 480, v.RGB)),
    ("UD_FORCE_GATHER", (1,), lambda v, g: ud(v, g, 1280, 720, 640, 360, v.YUV444)),
    ("ROTATE_NO_TILE", (1, 2), lambda v, g: rotate(v, g, 640, 360, 90.0)),
    ("ROTATE_NO_TILE", (2,), lambda v, g: rotate(v, g, 1000, 600, 270.0)),
    ("ROTATE_NO_TILE", (1,), lambda v, g: rotate(v, g, 640, 360, 180.0)),
    # any other angle: 0 the LDS-staged form (round 6), 1 the per-pixel gather form, 2..5 other tile shapes
    ("ROTATE_AFFINE", (1, 2, 3, 4, 5, 6), lambda v, g: rotate(v, g, 640, 360, 33.0)),
    ("ROTATE_AFFINE", (1, 3), lambda v, g: rotate(v, g, 1000, 600, -117.5)),
    # rows per wave (small launches pick 2 or 4 by themselves; batches 8)
    ("ROWS_PER_WAVE", (2, 4, 8), lambda v, g: ud(v, g, 1280, 720, 640, 358, v.RGB)),           # exact-2x kernel
    ("ROWS_PER_WAVE", (2, 4, 8), lambda v, g: ud(v, g, 1280, 720, 1280, 717, v.YUV444)),       # 1:1 kernel
    ("ROWS_PER_WAVE", (2, 4, 8), lambda v, g: ud(v, g, 1280, 720, 854, 477, v.RGB_PLANAR)),    # any-ratio kernel, staged
    ("ROWS_PER_WAVE", (2, 4, 8), lambda v, g: ud(v, g, 3000, 200, 300, 61, v.RGB)),            # any-ratio kernel, gather
    ("ROWS_PER_WAVE", (2, 4, 8), lambda v, g: resize(v, g, 1280, 720, 854, 478, v.Interpolation.LINEAR)),
    ("ROWS_PER_WAVE", (2, 4, 8), lambda v, g: resize(v, g, 1920, 1080, 480, 270, v.Interpolation.LINEAR)),   # point form, 4x
    ("ROWS_PER_WAVE", (2, 4, 8), lambda v, g: preproc(v, g, 1280, 720, 640, 382)),
    ("ROWS_PER_WAVE", (2, 4, 8), lambda v, g: preproc(v, g, 640, 362, 640, 362)),
    # how the blocking Run forms wait: completion word + spin (0) or hipStreamSynchronize (1)
    ("BLOCKING_WAIT", (1,), lambda v, g: nv12_rgb(v, g, 1920, 1080, v.RGB)),
    ("BLOCKING_WAIT", (1,), lambda v, g: ud(v, g, 1280, 720, 854, 480, v.RGB)),
]


@pytest.mark.parametrize("case", range(len(CASES)), ids=lambda i: f"{CASES[i][0]}-{i}")
def test_switch_cannot_change_the_result(vali, gpu, case):
    name, values, fn = CASES[case]
    default = vali.tuning.Get(name)
    want = fn(vali, gpu)
    for v in values:
        with vali.tuning.Override(**{name: v}):
            assert vali.tuning.Get(name) == v
            got = fn(vali, gpu)
        assert np.array_equal(got, want), (name, v)
    assert vali.tuning.Get(name) == default


def test_tracing_on_leaves_results_alone(vali, gpu):
    want = nv12_rgb(vali, gpu, 640, 360, vali.RGB)
    with vali.tuning.Override(ROCTX=1):
        assert vali.tuning.Get("ROCTX") == 1
        assert np.array_equal(nv12_rgb(vali, gpu, 640, 360, vali.RGB), want)
        assert np.array_equal(ud(vali, gpu, 640, 360, 320, 180, vali.RGB), ud(vali, gpu, 640, 360, 320, 180, vali.RGB))
    assert vali.tuning.Get("ROCTX") == 0


def test_null_stream_is_the_legacy_default_stream_of_the_tasks_gpu(vali, gpu, oracle):
    """PySurfaceConverter(gpu_id, 0) runs on stream 0 as handed over (the reference uses the stream it is given,
    PySurfaceConverter.cpp:33-45; torch's default stream is 0, and a caller who passes it expects ordering with their own
    work there -- ADVICE r02); the device a null stream does not carry is the task's gpu_id, made current before every call.
    No stream at all = the resource manager's stream."""
    import torch
    cvt = vali.PySurfaceConverter(gpu, 0)
    assert cvt.Stream == 0
    assert vali.PySurfaceConverter(gpu).Stream == vali.HipResMgr.Instance().GetStream(gpu) != 0
    w, h = 640, 360
    rng = np.random.default_rng(5)
    nv12 = rng.integers(0, 256, (h * 3 // 2, w), dtype=np.uint8)
    # the source is WRITTEN on torch's default stream (0) right before the conversion reads it, the result is read by
    # torch on stream 0 right after: no synchronisation anywhere in between
    t_src = torch.zeros((h * 3 // 2, w), dtype=torch.uint8, device=f"cuda:{gpu}")
    t_dst = torch.zeros((h, w * 3), dtype=torch.uint8, device=f"cuda:{gpu}")
    src = vali.Surface.from_dlpack(t_src, vali.NV12)
    dst = vali.Surface.from_dlpack(t_dst, vali.RGB)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    host = torch.from_numpy(nv12).pin_memory()
    for _ in range(3):
        t_src.copy_(host, non_blocking=True)
        assert cvt.RunAsync(src, dst, cc) == (True, vali.TaskExecInfo.SUCCESS)
        got = t_dst.cpu().numpy()
    from vali_amd import tasks
    want = oracle.nv12_to_rgb(nv12, w, h, oracle.csc_from_tuple(tasks.CSC_NPP_709CSC), "RGB")
    assert np.array_equal(got.reshape(-1), want.reshape(-1))
    assert cvt.Run(src, dst, cc)[0] and vali.PySurfaceResizer(vali.NV12, gpu, 0).Stream == 0


@pytest.mark.parametrize("mode", [0, 1])
def test_blocking_run_returns_after_the_work_not_before(vali, gpu, oracle, mode):
    """Run = RunAsync + host wait (PySurfaceConverter.cpp:76-86).  The wait is a completion word the stream writes into
    pinned host memory (vali_stream_wait): a long launch (a batch of 2160p frames: past the 150 us of spinning, into the
    hipStreamSynchronize fallback) and many short ones must all be finished when the call returns -- the result is read
    back through a DIFFERENT stream right after, with no other synchronisation."""
    from vali_amd._native import shim
    w, h, n = 3840, 2160, 24
    rng = np.random.default_rng(8)
    nv12 = rng.integers(0, 256, (h * 3 // 2, w), dtype=np.uint8)
    srcs = [vali.Surface.Make(vali.NV12, w, h, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(vali.RGB, w, h, gpu) for _ in range(n)]
    for s in srcs:
        assert vali.PyFrameUploader(gpu).Run(nv12.reshape(-1), s)[0]
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    from vali_amd import tasks
    want = oracle.nv12_to_rgb(nv12, w, h, oracle.csc_from_tuple(tasks.CSC_NPP_709CSC), "RGB").reshape(-1)
    other = shim.stream_create(gpu)
    down = vali.PySurfaceDownloader(gpu, other)
    with vali.tuning.Override(BLOCKING_WAIT=mode):
        cvt = vali.PySurfaceConverter(gpu)
        for d in dsts:
            shim.memset2d_async(gpu, d.Planes[0].GpuMem, d.Planes[0].Pitch, 0, w * 3, h, cvt.Stream)
        assert cvt.RunBatch(srcs, dsts, cc) == (True, vali.TaskExecInfo.SUCCESS)
        got = np.zeros(dsts[-1].HostSize, np.uint8)
        assert down.Run(dsts[-1], got)[0] and np.array_equal(got, want)
        small_s, small_d = vali.Surface.Make(vali.NV12, 64, 48, gpu), vali.Surface.Make(vali.RGB, 64, 48, gpu)
        tiny = rng.integers(0, 256, (72, 64), dtype=np.uint8)
        tiny_want = oracle.nv12_to_rgb(tiny, 64, 48, oracle.csc_from_tuple(tasks.CSC_NPP_709CSC), "RGB").reshape(-1)
        for i in range(200):
            assert vali.PyFrameUploader(gpu, cvt.Stream).Run(np.roll(tiny.reshape(-1), 0), small_s)[0]
            shim.memset2d_async(gpu, small_d.Planes[0].GpuMem, small_d.Planes[0].Pitch, i & 0xff, 64 * 3, 48, cvt.Stream)
            assert cvt.Run(small_s, small_d, cc)[0]
            out = np.zeros(small_d.HostSize, np.uint8)
            assert down.Run(small_d, out)[0] and np.array_equal(out, tiny_want)
    shim.stream_destroy(gpu, other)
