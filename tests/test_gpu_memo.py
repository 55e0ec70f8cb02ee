"""Every single-surface task keeps a small memo of resolved C calls keyed by the surfaces' descriptor
objects (vali_amd/tasks.py `_SurfaceTask._memo`).  Whatever sequence of calls a caller makes, the
result must be what a fresh task (no memo) produces: repeated pairs, alternating pairs, changed
arguments on the same pair, and a surface that was re-pointed to other memory."""
import numpy as np
import pytest

from conftest import make_nv12

pytestmark = pytest.mark.gpu


def get(vali, gpu, surf):
    out = np.zeros(surf.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(surf, out)[0]
    return out


def put(vali, gpu, fmt, w, h, host):
    s = vali.Surface.Make(fmt, w, h, gpu)
    assert vali.PyFrameUploader(gpu).Run(np.ascontiguousarray(host).reshape(-1).view(np.uint8), s)[0]
    return s


def sync(vali, gpu, task):
    ev = vali.CudaStreamEvent(task.Stream, gpu)
    ev.Record()
    ev.Wait()


def test_resizer_ud_rotator_preprocessor_memo(vali, gpu):
    w, h = 320, 180
    nvs = [put(vali, gpu, vali.NV12, w, h, make_nv12(w, h, 70 + i)) for i in range(2)]
    rng = np.random.default_rng(5)
    rgbs = [put(vali, gpu, vali.RGB, w, h, rng.integers(0, 256, w * h * 3, dtype=np.uint8)) for _ in range(2)]

    cases = []   # (name, make task, call(task, src, dst) -> result, srcs, make dst)
    cases.append(("resize", lambda: vali.PySurfaceResizer(vali.NV12, gpu),
                  lambda t, s, d: t.RunAsync(s, d), nvs, lambda: vali.Surface.Make(vali.NV12, 200, 120, gpu)))
    cases.append(("resize-lanczos", lambda: vali.PySurfaceResizer(vali.NV12, gpu, interpolation=vali.Interpolation.LANCZOS),
                  lambda t, s, d: t.RunAsync(s, d), nvs, lambda: vali.Surface.Make(vali.NV12, 200, 120, gpu)))
    cases.append(("ud", lambda: vali.PySurfaceUD(gpu),
                  lambda t, s, d: t.RunAsync(s, d), nvs, lambda: vali.Surface.Make(vali.RGB, 160, 90, gpu)))
    cases.append(("preproc", lambda: vali.PySurfacePreprocessor(gpu, mean=(0.4, 0.5, 0.6), std=(0.2, 0.3, 0.4), div=255.0),
                  lambda t, s, d: t.RunAsync(s, d), nvs, lambda: vali.Surface.Make(vali.RGB_32F_PLANAR, 128, 96, gpu)))
    for ang in (90.0, 30.0):
        cases.append((f"rotate{ang}", lambda: vali.PySurfaceRotator(gpu),
                      lambda t, s, d, a=ang: t.RunAsync(s, d, a), rgbs,
                      (lambda: vali.Surface.Make(vali.RGB, h, w, gpu)) if ang == 90.0 else (lambda: vali.Surface.Make(vali.RGB, w, h, gpu))))
    for name, mk, call, srcs, mkdst0 in cases:
        def mkdst(mk0=mkdst0):     # zero-filled: an arbitrary-angle rotation leaves the pixels it misses untouched
            d = mk0()
            assert vali.PyFrameUploader(gpu).Run(np.zeros(d.HostSize, np.uint8), d)[0]
            return d
        task = mk()
        dsts = [mkdst(), mkdst()]
        want = []
        for s_ in srcs:                                      # fresh task per call: never memoised
            d = mkdst()
            t = mk()
            assert call(t, s_, d) == (True, vali.TaskExecInfo.SUCCESS), name
            sync(vali, gpu, t)
            want.append(get(vali, gpu, d))
        for si, di in ((0, 0), (0, 0), (0, 0), (1, 1), (0, 1), (1, 0), (1, 0), (0, 0)):
            assert vali.PyFrameUploader(gpu).Run(np.zeros(dsts[di].HostSize, np.uint8), dsts[di])[0]
            assert call(task, srcs[si], dsts[di]) == (True, vali.TaskExecInfo.SUCCESS), name
            sync(vali, gpu, task)
            assert np.array_equal(get(vali, gpu, dsts[di]), want[si]), (name, si, di)


def test_rotator_memo_distinguishes_angles_and_shifts(vali, gpu):
    w, h = 256, 128
    rng = np.random.default_rng(6)
    src = put(vali, gpu, vali.RGB, w, h, rng.integers(0, 256, w * h * 3, dtype=np.uint8))
    dst = vali.Surface.Make(vali.RGB, w, h, gpu)
    rot = vali.PySurfaceRotator(gpu)
    outs = {}
    for args in ((30.0, 0.0, 0.0), (31.0, 0.0, 0.0), (30.0, 5.0, 0.0), (30.0, 0.0, -7.0), (180.0, 0.0, 0.0), (30.0, 0.0, 0.0)):
        assert vali.PyFrameUploader(gpu).Run(np.zeros(dst.HostSize, np.uint8), dst)[0]
        for _ in range(2):
            assert rot.RunAsync(src, dst, *args)[0]
        sync(vali, gpu, rot)
        got = get(vali, gpu, dst)
        fresh = vali.Surface.Make(vali.RGB, w, h, gpu)
        assert vali.PyFrameUploader(gpu).Run(np.zeros(fresh.HostSize, np.uint8), fresh)[0]
        assert vali.PySurfaceRotator(gpu).Run(src, fresh, *args)[0]
        assert np.array_equal(got, get(vali, gpu, fresh)), args
        outs[args] = got
    assert not np.array_equal(outs[(30.0, 0.0, 0.0)], outs[(31.0, 0.0, 0.0)])
    assert not np.array_equal(outs[(30.0, 0.0, 0.0)], outs[(30.0, 5.0, 0.0)])


def test_memo_follows_a_repointed_surface(vali, gpu, oracle):
    """a borrowed (DLPack) surface re-created over other memory has a new descriptor: no stale hit"""
    import torch
    w, h = 128, 64
    frames = [make_nv12(w, h, 80 + i) for i in range(2)]
    bufs = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in frames]
    rs = vali.PySurfaceResizer(vali.NV12, gpu)
    dst = vali.Surface.Make(vali.NV12, 64, 32, gpu)
    for i in (0, 1, 0, 1):
        src = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(bufs[i]), vali.NV12)
        for _ in range(2):
            assert rs.RunAsync(src, dst)[0]
        sync(vali, gpu, rs)
        assert np.array_equal(get(vali, gpu, dst), oracle.resize_surface(frames[i].reshape(-1), "NV12", w, h, 64, 32, "linear"))
    assert len(rs._memo) <= 16


def test_memo_is_bounded_and_does_not_keep_surfaces_alive(vali, gpu):
    import gc
    import weakref
    cv = vali.PySurfaceConverter(gpu)
    nv = vali.Surface.Make(vali.NV12, 64, 32, gpu)
    refs = []
    for _ in range(40):
        d = vali.Surface.Make(vali.RGB, 64, 32, gpu)
        assert cv.RunAsync(nv, d)[0]
        refs.append(weakref.ref(d))
        del d
    sync(vali, gpu, cv)
    gc.collect()
    assert len(cv._memo) <= 16
    assert all(r() is None for r in refs)


def test_memo_sees_a_mutated_colour_context(vali, gpu):
    """ColorspaceConversionContext is a mutable struct (as in the reference): the memo keys on its values"""
    w, h = 128, 64
    src = put(vali, gpu, vali.NV12, w, h, make_nv12(w, h, 3))
    dst = vali.Surface.Make(vali.RGB, w, h, gpu)
    cv = vali.PySurfaceConverter(gpu)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    outs = []
    for space, rng_ in ((vali.ColorSpace.BT_709, vali.ColorRange.MPEG), (vali.ColorSpace.BT_601, vali.ColorRange.JPEG),
                        (vali.ColorSpace.BT_709, vali.ColorRange.JPEG), (vali.ColorSpace.BT_709, vali.ColorRange.MPEG)):
        cc.color_space, cc.color_range = space, rng_
        for _ in range(2):
            assert cv.RunAsync(src, dst, cc)[0]
        sync(vali, gpu, cv)
        got = get(vali, gpu, dst)
        fresh = vali.Surface.Make(vali.RGB, w, h, gpu)
        assert vali.PySurfaceConverter(gpu).Run(src, fresh, vali.ColorspaceConversionContext(space, rng_))[0]
        assert np.array_equal(got, get(vali, gpu, fresh)), (space, rng_)
        outs.append(got)
    assert not np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[3])


def test_list_form_of_run_batch_uploads_its_descriptors_once(vali, gpu):
    """RunBatchAsync(srcs, dsts): the descriptor arrays are uploaded on the first call and kept with the task
    (a temporary SurfaceBatch per call allocated, synchronised and freed around every launch)."""
    w, h, n = 64, 48, 3
    cvt = vali.PySurfaceConverter(gpu)
    srcs = [vali.Surface.Make(vali.NV12, w, h, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(vali.RGB, w, h, gpu) for _ in range(n)]
    for _ in range(4):
        assert cvt.RunBatchAsync(srcs, dsts) == (True, vali.TaskExecInfo.SUCCESS)
    assert len(cvt._batches) == 1
    b = next(iter(cvt._batches.values()))
    assert cvt.RunBatch(list(srcs), list(dsts))[0] and next(iter(cvt._batches.values())) is b      # equal lists: the same batch
    other = [vali.Surface.Make(vali.RGB, w, h, gpu) for _ in range(n)]
    assert cvt.RunBatch(srcs, other)[0] and len(cvt._batches) == 2
    for _ in range(10):                                                                              # bounded
        assert cvt.RunBatch(srcs, [vali.Surface.Make(vali.RGB, w, h, gpu) for _ in range(n)])[0]
    assert len(cvt._batches) <= 8
    with pytest.raises(ValueError):
        cvt.RunBatch(srcs)


def test_surface_batch_cannot_be_created_while_capturing(vali, gpu):
    from vali_amd._native import shim

    stream = shim.stream_create(gpu)
    cvt = vali.PySurfaceConverter(gpu, stream)
    srcs = [vali.Surface.Make(vali.NV12, 64, 48, gpu) for _ in range(2)]
    dsts = [vali.Surface.Make(vali.RGB, 64, 48, gpu) for _ in range(2)]
    ready = cvt.PrepareBatch(srcs, dsts)
    fresh = [vali.Surface.Make(vali.RGB, 64, 48, gpu) for _ in range(2)]
    cap = vali.StreamCapture(stream, gpu)
    with cap:
        with pytest.raises(RuntimeError, match="capturing"):
            cvt.RunBatchAsync(srcs, fresh)
        assert cvt.RunBatchAsync(ready)[0]              # a prepared batch records fine
    cap.Keep(ready)
    cap.Launch()
    shim.stream_sync(gpu, stream)


def test_batch_cache_does_not_pin_surfaces(vali, gpu):
    """the task's descriptor-array cache keeps no reference to the caller's surfaces: dropping the lists frees them"""
    import gc
    import weakref

    cvt = vali.PySurfaceConverter(gpu)
    srcs = [vali.Surface.Make(vali.NV12, 64, 48, gpu) for _ in range(2)]
    dsts = [vali.Surface.Make(vali.RGB, 64, 48, gpu) for _ in range(2)]
    assert cvt.RunBatch(srcs, dsts)[0]
    ref = weakref.ref(dsts[0])
    del srcs, dsts
    gc.collect()
    assert ref() is None
    kept = cvt.PrepareBatch([vali.Surface.Make(vali.NV12, 64, 48, gpu)], [vali.Surface.Make(vali.RGB, 64, 48, gpu)])
    gc.collect()
    assert cvt.RunBatch(kept)[0]          # a PREPARED batch owns its surfaces
