"""Surface / SurfacePlane behaviour on the device.

Mirrors reference tests/test_PySurface.py (:77-96 plane DLPack, :137-166 surface DLPack,
:168-197 from_dlpack, :293-346 Make for all formats) and tests/test_GpuMem.py:58-62."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALL = ["Y", "RGB", "NV12", "YUV420", "RGB_PLANAR", "BGR", "YUV444", "YUV444_10bit",
       "YUV420_10bit", "RGB_32F", "RGB_32F_PLANAR", "YUV422", "P10", "P12"]
ONE_PLANE = {"Y", "RGB", "BGR", "RGB_32F", "NV12", "P10", "P12", "RGB_PLANAR", "RGB_32F_PLANAR"}


@pytest.mark.parametrize("name", ALL)
def test_surface_make_all_formats(vali, gpu, name):
    fmt = vali.PixelFormat[name]
    s = vali.Surface.Make(fmt, 1920, 1080, gpu)
    assert (s.Width, s.Height, s.Format) == (1920, 1080, fmt)
    assert s.HostSize > 0 and not s.IsEmpty and s.IsOwnMemory
    assert s.NumPlanes == (1 if name in ONE_PLANE else 3)
    for p in s.Planes:                                   # test_GpuMem.py:58-62
        assert p.GpuMem == p.__cuda_array_interface__["data"][0]
        assert p.Pitch >= p.Width * p.ElemSize and p.Pitch % 256 == 0


def test_upload_download_roundtrip(vali, gpu):
    for name in ("NV12", "YUV420", "RGB", "RGB_32F_PLANAR", "P10"):
        s = vali.Surface.Make(vali.PixelFormat[name], 130, 66, gpu)
        src = np.random.default_rng(1).integers(0, 256, s.HostSize, dtype=np.uint8)
        assert vali.PyFrameUploader(gpu).Run(src, s) == (True, vali.TaskExecInfo.SUCCESS)
        dst = np.zeros_like(src)
        assert vali.PySurfaceDownloader(gpu).Run(s, dst) == (True, vali.TaskExecInfo.SUCCESS)
        assert np.array_equal(src, dst)
        ok, info = vali.PyFrameUploader(gpu).Run(src[:-1], s)
        assert not ok and info == vali.TaskExecInfo.SRC_DST_SIZE_MISMATCH
        c = s.Clone()
        dst2 = np.zeros_like(src)
        assert vali.PySurfaceDownloader(gpu).Run(c, dst2)[0] and np.array_equal(src, dst2)


def test_dlpack_export_plane_and_surface(vali, gpu):
    import torch

    w, h = 200, 100
    s = vali.Surface.Make(vali.RGB, w, h, gpu)
    src = np.random.default_rng(2).integers(0, 256, s.HostSize, dtype=np.uint8)
    assert vali.PyFrameUploader(gpu).Run(src, s)[0]
    t_plane = torch.from_dlpack(s.Planes[0])
    assert tuple(t_plane.shape) == (h, 3 * w) and t_plane.shape[0] * t_plane.shape[1] == s.HostSize
    assert np.array_equal(t_plane.cpu().numpy().ravel(), src)
    t = torch.from_dlpack(s)
    assert tuple(t.shape) == (h, w, 3) and t.is_cuda
    assert np.array_equal(t.cpu().numpy().ravel(), src)
    assert s.__dlpack_device__() == (10, gpu)            # kDLROCM
    p = vali.Surface.Make(vali.RGB_32F_PLANAR, w, h, gpu)
    tp = torch.from_dlpack(p)
    assert tuple(tp.shape) == (3, h, w) and tp.dtype == torch.float32
    with pytest.raises(RuntimeError):
        vali.Surface.Make(vali.YUV420, w, h, gpu).__dlpack__()


def test_surface_from_tensor(vali, gpu):
    import torch

    w, h = 96, 40
    rgb = np.random.default_rng(3).integers(0, 256, (h, w * 3), dtype=np.uint8)
    t = torch.from_numpy(rgb).to("cuda")
    s = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(t))
    assert not s.IsEmpty and not s.IsOwnMemory and s.Format == vali.RGB
    assert (s.Width, s.Height, s.HostSize) == (w, h, rgb.size)
    out = np.zeros(s.HostSize, np.uint8)
    torch.cuda.synchronize()
    assert vali.PySurfaceDownloader(gpu).Run(s, out)[0]
    assert np.array_equal(out, rgb.ravel())
    with pytest.raises(RuntimeError):
        s.__dlpack_device__()                            # PySurface.cpp:389-393
    s2 = vali.Surface.from_cai(t, vali.Y)
    assert (s2.Width, s2.Height) == (w * 3, h)


def test_empty_surfaces_are_rejected_not_crashed(vali, gpu):
    """Surface.Make(format) alone gives an empty surface (MemoryInterfaces.cpp:336-367); every
    task must answer with an error code."""
    T = vali.TaskExecInfo
    e_nv12, e_rgb = vali.Surface.Make(vali.NV12), vali.Surface.Make(vali.RGB)
    assert e_nv12.IsEmpty and e_nv12.HostSize == 0 and e_nv12.Width == 0
    full = vali.Surface.Make(vali.RGB, 64, 48, gpu)
    assert vali.PySurfaceConverter(gpu).Run(e_nv12, full) == (False, T.INVALID_INPUT)
    assert vali.PySurfaceConverter(gpu).Run(e_nv12, e_rgb) == (False, T.INVALID_INPUT)
    assert vali.PySurfaceResizer(vali.RGB, gpu).Run(e_rgb, full) == (False, T.INVALID_INPUT)
    assert vali.PySurfaceRotator(gpu).Run(e_rgb, full, 90.0) == (False, T.INVALID_INPUT)
    assert vali.PySurfaceUD(gpu).Run(e_nv12, full) == (False, T.INVALID_INPUT)
    assert vali.PyFrameUploader(gpu).Run(np.zeros(16, np.uint8), e_rgb) == (False, T.INVALID_INPUT)
    assert vali.PySurfaceDownloader(gpu).Run(e_rgb, np.zeros(16, np.uint8)) == (False, T.INVALID_INPUT)
    assert e_rgb.Clone().IsEmpty


def test_streams_and_events(vali, gpu):
    """Tasks sharing a stream are stream-ordered; CudaStreamEvent Record/Wait synchronises."""
    a = vali.PySurfaceConverter(gpu)
    b = vali.PySurfaceConverter(gpu, a.Stream)
    assert a.Stream == b.Stream and a.Stream != 0
    assert vali.PySurfaceResizer(vali.NV12, gpu).Stream == a.Stream      # per-GPU default stream
    ev = vali.CudaStreamEvent(a.Stream, gpu)
    ev.Record()
    ev.Wait()
    assert vali.GetNumGpus() >= 1


def test_cuda_buffer(vali, gpu):
    """CudaBuffer (reference VALI.cpp:349-441): Make / props / Clone / CopyFrom semantics."""
    import torch

    b = vali.CudaBuffer.Make(4, 1000, gpu)
    assert (b.ElemSize, b.NumElems, b.RawMemSize) == (4, 1000, 4000) and b.GpuMem != 0
    # fill through a foreign view of the same memory, then Clone / CopyFrom move the bytes
    src = torch.arange(1000, dtype=torch.int32, device="cuda")
    shim = __import__("vali_amd._native", fromlist=["shim"]).shim
    shim.memcpy2d_async(gpu, b.GpuMem, 4000, src.data_ptr(), 4000, 4000, 1, 2, 0)
    shim.stream_sync(gpu, 0)
    c = b.Clone()
    assert c.GpuMem != b.GpuMem and c.RawMemSize == 4000
    d = vali.CudaBuffer.Make(1, 4000, gpu)                      # same raw size, other element size
    d.CopyFrom(c, gpu_id=gpu)
    out = torch.empty(1000, dtype=torch.int32, device="cuda")
    shim.memcpy2d_async(gpu, out.data_ptr(), 4000, d.GpuMem, 4000, 4000, 1, 2, 0)
    shim.stream_sync(gpu, 0)
    assert torch.equal(out, src)
    with pytest.raises(RuntimeError):                           # VALI.cpp:31-33
        vali.CudaBuffer.Make(4, 999, gpu).CopyFrom(b, gpu_id=gpu)
    with pytest.raises(TypeError):
        vali.CudaBuffer()


def test_two_threads_two_streams(vali, gpu, oracle):
    """one task instance = one stream, no internal locking, distinct instances may
    run from distinct threads concurrently (every Run releases the GIL)."""
    import threading

    from conftest import make_nv12

    w, h, iters = 640, 360, 40
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    results, errors = {}, []

    def worker(idx):
        try:
            nv = make_nv12(w, h, 100 + idx)
            stream = vali._native.shim.stream_create(gpu)
            cvt = vali.PySurfaceConverter(gpu, stream)
            rs = vali.PySurfaceResizer(vali.NV12, gpu, stream)
            src = vali.Surface.Make(vali.NV12, w, h, gpu)
            half = vali.Surface.Make(vali.NV12, w // 2, h // 2, gpu)
            dst = vali.Surface.Make(vali.RGB, w // 2, h // 2, gpu)
            assert vali.PyFrameUploader(gpu, stream).Run(nv.reshape(-1), src)[0]
            for _ in range(iters):
                assert rs.RunAsync(src, half)[0] and cvt.RunAsync(half, dst, cc)[0]
            assert cvt.Run(half, dst, cc)[0]
            out = np.zeros(dst.HostSize, np.uint8)
            assert vali.PySurfaceDownloader(gpu, stream).Run(dst, out)[0]
            small = oracle.resize_surface(nv.reshape(-1), "NV12", w, h, w // 2, h // 2).reshape(h * 3 // 4, w // 2)
            results[idx] = np.array_equal(out, oracle.nv12_to_rgb(small, w // 2, h // 2, oracle.csc(1), "RGB").reshape(-1))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    assert results == {0: True, 1: True, 2: True, 3: True}


def test_threads_blocking_on_one_shared_stream(vali, gpu, oracle):
    """Tasks built WITHOUT a stream share their GPU's manager stream; their blocking Run forms wait on ONE completion
    word whose values two threads take concurrently (the GIL is released around the wait): the counter and the
    enqueue of its write are one locked step (runtime.hip vali_stream_wait)."""
    import threading

    from conftest import make_nv12

    w, h, iters = 320, 180, 300
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    errors, results = [], {}

    def worker(idx):
        try:
            nv = make_nv12(w, h, 300 + idx)
            cvt = vali.PySurfaceConverter(gpu)                  # the manager's stream, shared by all four threads
            src = vali.Surface.Make(vali.NV12, w, h, gpu)
            dst = vali.Surface.Make(vali.RGB, w, h, gpu)
            assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), src)[0]
            want = oracle.nv12_to_rgb(nv, w, h, oracle.csc(1), "RGB").reshape(-1)
            ok = True
            for _ in range(iters):
                assert cvt.Run(src, dst, cc)[0]                 # blocking: returns only when the kernel has finished
            out = np.zeros(dst.HostSize, np.uint8)
            assert vali.PySurfaceDownloader(gpu).Run(dst, out)[0]
            results[idx] = ok and np.array_equal(out, want)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    assert results == {0: True, 1: True, 2: True, 3: True}


def test_null_stream_task_restores_the_callers_device(vali, gpu, oracle):
    """A task on (gpu, stream 0) makes its GPU current for the call and puts the caller's device back (the pop of the
    reference's CudaCtxPush, CudaUtils.hpp:77-90); its wrappers hold no strong reference to the task."""
    import gc
    import weakref

    from conftest import make_nv12
    shim = vali._native.shim
    assert shim.device_get() == gpu or shim.device_get() >= 0
    before = shim.device_get()
    w, h = 128, 64
    nv = make_nv12(w, h, 5)
    cvt = vali.PySurfaceConverter(gpu, 0)
    src, dst = vali.Surface.Make(vali.NV12, w, h, gpu), vali.Surface.Make(vali.RGB, w, h, gpu)
    assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), src)[0]
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    assert cvt.Run(src, dst, cc) == (True, vali.TaskExecInfo.SUCCESS) and cvt.Stream == 0
    assert shim.device_get() == before
    out = np.zeros(dst.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(dst, out)[0]
    assert np.array_equal(out, oracle.nv12_to_rgb(nv, w, h, oracle.csc(1), "RGB").reshape(-1))
    gc.disable()
    try:
        ref = weakref.ref(cvt)
        del cvt
        assert ref() is None                                    # reference counting alone frees it: no cycle
    finally:
        gc.enable()


def test_stream_capture_replays_a_chain(vali, gpu, oracle):
    """StreamCapture: resize -> convert -> planar recorded once, replayed per frame; results are
    the eager results, for every refill of the captured input surface."""
    from conftest import make_nv12

    w, h = 1280, 720
    stream = vali._native.shim.stream_create(gpu)
    rs, cvt = vali.PySurfaceResizer(vali.NV12, gpu, stream), vali.PySurfaceConverter(gpu, stream)
    up, dn = vali.PyFrameUploader(gpu, stream), vali.PySurfaceDownloader(gpu, stream)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    src = vali.Surface.Make(vali.NV12, w, h, gpu)
    half = vali.Surface.Make(vali.NV12, w // 2, h // 2, gpu)
    rgb = vali.Surface.Make(vali.RGB, w // 2, h // 2, gpu)
    pl = vali.Surface.Make(vali.RGB_PLANAR, w // 2, h // 2, gpu)
    # warm every kernel once outside the capture (module loading is not capturable)
    assert rs.Run(src, half)[0] and cvt.Run(half, rgb, cc)[0] and cvt.Run(rgb, pl)[0]
    cap = vali.StreamCapture(stream, gpu).Keep(src, half, rgb, pl)
    with cap:
        assert rs.RunAsync(src, half)[0]
        assert cvt.RunAsync(half, rgb, cc)[0]
        assert cvt.RunAsync(rgb, pl)[0]
    for seed in (1, 2, 3):
        nv = make_nv12(w, h, seed)
        assert up.Run(nv.reshape(-1), src)[0]
        vali._native.shim.memset2d_async(gpu, pl._planes[0].GpuMem, pl._planes[0].Pitch, 0, w // 2, 3 * h // 2, stream)
        cap.Launch()
        out = np.zeros(pl.HostSize, np.uint8)
        assert dn.Run(pl, out)[0]
        small = oracle.resize_surface(nv.reshape(-1), "NV12", w, h, w // 2, h // 2).reshape(h * 3 // 4, w // 2)
        want = oracle.nv12_to_rgb(small, w // 2, h // 2, oracle.csc(1), "RGB_PLANAR").reshape(-1)
        assert np.array_equal(out, want)
    with pytest.raises(RuntimeError):
        vali.StreamCapture(stream, gpu).Launch()


def test_no_device_memory_leak_over_many_surfaces_and_tasks(vali, gpu):
    """Surfaces, batches, tasks, events and captures give their device memory back: 300 rounds of
    allocate / run / drop leave the free-memory figure where it started (within 64 MiB)."""
    import gc

    import torch

    def free_bytes():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info(gpu)[0]

    def round_trip():
        srcs = [vali.Surface.Make(vali.NV12, 1920, 1080, gpu) for _ in range(4)]
        dsts = [vali.Surface.Make(vali.RGB, 1920, 1080, gpu) for _ in range(4)]
        cvt = vali.PySurfaceConverter(gpu)
        assert cvt.RunBatch(srcs, dsts)[0]
        rs = vali.PySurfaceResizer(vali.NV12, gpu)
        small = vali.Surface.Make(vali.NV12, 640, 360, gpu)
        assert rs.Run(srcs[0], small)[0]
        c = srcs[0].Clone()
        b = vali.CudaBuffer.Make(1, 1 << 20, gpu)
        del srcs, dsts, cvt, rs, small, c, b

    round_trip()
    gc.collect()
    before = free_bytes()
    for _ in range(300):
        round_trip()
    gc.collect()
    after = free_bytes()
    assert before - after < 64 << 20, (before, after)


def test_batch_argument_errors(vali, gpu):
    """RunBatch: empty lists, mismatched lengths / sizes / formats come back as errors, not crashes."""
    cvt = vali.PySurfaceConverter(gpu)
    a = [vali.Surface.Make(vali.NV12, 64, 48, gpu) for _ in range(2)]
    b = [vali.Surface.Make(vali.RGB, 64, 48, gpu) for _ in range(2)]
    with pytest.raises((ValueError, RuntimeError)):
        cvt.RunBatch([], [])
    with pytest.raises((ValueError, RuntimeError)):
        cvt.RunBatch(a, b[:1])
    with pytest.raises((ValueError, RuntimeError)):
        cvt.RunBatch(a, [b[0], vali.Surface.Make(vali.RGB, 32, 48, gpu)])
    with pytest.raises((ValueError, RuntimeError)):
        cvt.RunBatch([a[0], vali.Surface.Make(vali.YUV420, 64, 48, gpu)], b)
    wrong = [vali.Surface.Make(vali.RGB, 32, 24, gpu) for _ in range(2)]
    assert cvt.RunBatch(a, wrong) == (False, vali.TaskExecInfo.INVALID_INPUT)
    assert cvt.RunBatch(a, b) == (True, vali.TaskExecInfo.SUCCESS)
