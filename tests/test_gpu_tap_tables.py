"""Tap tables of the Lanczos / bicubic resizers (vali_amd/csrc/tap_table.hip, round 6): one arena per device reserved by the
first call, least-recently-used eviction instead of refusal, no allocation on the hot path, fall-backs visible in a counter.
Every result stays bit-exact against the oracle whatever the table traffic."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _up(vali, gpu, fmt, w, h, host):
    s = vali.Surface.Make(fmt, w, h, gpu)
    assert vali.PyFrameUploader(gpu).Run(np.ascontiguousarray(host).reshape(-1).view(np.uint8), s)[0]
    return s


def _down(vali, gpu, s):
    out = np.zeros(s.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu).Run(s, out)[0]
    return out


def test_eviction_instead_of_refusal_and_results_stay_exact(vali, gpu, oracle):
    """More geometries than the table count allows: old tables are evicted (counter), none is refused (counter stays 0), and
    a geometry that comes BACK after its table was evicted is rebuilt and gives the same bytes."""
    rng = np.random.default_rng(5)
    sw, sh = 640, 360
    host = rng.integers(0, 256, sw * sh, dtype=np.uint8)
    src = _up(vali, gpu, vali.Y, sw, sh, host)
    rs = vali.PySurfaceResizer(vali.Y, gpu)          # Lanczos, the reference's default
    geoms = sorted({(int(w) // 2 * 2, int(h) // 2 * 2) for w, h in zip(rng.integers(330, 630, 40), rng.integers(190, 350, 40))})
    with vali.tuning.Override(TAP_MAX_TABLES=8, TAP_FALLBACKS=0, TAP_EVICTIONS=0):
        first = {}
        for dw, dh in geoms + geoms[:6]:            # the first six come back after > 8 other tables
            d = vali.Surface.Make(vali.Y, dw, dh, gpu)
            assert rs.Run(src, d)[0]
            got = _down(vali, gpu, d)
            if (dw, dh) in first:
                assert np.array_equal(got, first[(dw, dh)])
            else:
                first[(dw, dh)] = got
                assert np.array_equal(got, oracle.resize_surface(host, "Y", sw, sh, dw, dh, "lanczos")), (dw, dh)
        assert vali.tuning.Get("TAP_FALLBACKS") == 0
        assert vali.tuning.Get("TAP_EVICTIONS") >= len(geoms) - 8


def test_eviction_across_streams_orders_the_rewrite_behind_the_readers(vali, gpu, oracle):
    """Two streams share the smallest table budget (8; one geometry = two axes): calls of one stream evict tables the other
    stream's queued launches may still be reading.  Every output must still be exact: the kernel that writes a new table into
    re-used space waits for what the evicted table's readers have been given."""
    from vali_amd._native import shim

    rng = np.random.default_rng(6)
    sw, sh = 1280, 720
    host = rng.integers(0, 256, sw * sh, dtype=np.uint8)
    src = _up(vali, gpu, vali.Y, sw, sh, host)
    sa, sb = shim.stream_create(gpu), shim.stream_create(gpu)
    ra, rb = vali.PySurfaceResizer(vali.Y, gpu, sa), vali.PySurfaceResizer(vali.Y, gpu, sb)
    ga = [(1000 - 2 * k, 560 - 2 * k) for k in range(16)]
    gb = [(900 - 2 * k, 500 - 2 * k) for k in range(16)]
    try:
        with vali.tuning.Override(TAP_MAX_TABLES=8, TAP_FALLBACKS=0, TAP_EVICTIONS=0):   # (8 is the least the library accepts)
            outs = []
            for (aw, ah), (bw, bh) in zip(ga, gb):
                da, db = vali.Surface.Make(vali.Y, aw, ah, gpu), vali.Surface.Make(vali.Y, bw, bh, gpu)
                assert ra.RunAsync(src, da)[0]
                assert rb.RunAsync(src, db)[0]
                outs += [(da, aw, ah), (db, bw, bh)]
            shim.stream_sync(gpu, sa)
            shim.stream_sync(gpu, sb)
            for d, w, h in outs:
                assert np.array_equal(_down(vali, gpu, d), oracle.resize_surface(host, "Y", sw, sh, w, h, "lanczos")), (w, h)
            assert vali.tuning.Get("TAP_FALLBACKS") == 0 and vali.tuning.Get("TAP_EVICTIONS") >= 40
    finally:
        shim.stream_sync(gpu, sa)
        shim.stream_sync(gpu, sb)
        del ra, rb
        shim.stream_destroy(gpu, sa)
        shim.stream_destroy(gpu, sb)


def test_a_new_geometry_does_not_wait_for_another_streams_work(vali, gpu):
    """After the arena exists, a first call with a new geometry allocates nothing: issued while another stream has tens of
    milliseconds of work queued, it returns to the host in a fraction of that (a hipMalloc there may wait for the device)."""
    from vali_amd._native import shim

    w, h, n = 3840, 2160, 48
    busy_stream, my_stream = shim.stream_create(gpu), shim.stream_create(gpu)
    try:
        cv = vali.PySurfaceConverter(gpu, busy_stream)
        srcs = [vali.Surface.Make(vali.NV12, w, h, gpu) for _ in range(n)]
        dsts = [vali.Surface.Make(vali.RGB, w, h, gpu) for _ in range(n)]
        cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
        batch = cv.PrepareBatch(srcs, dsts)
        rs = vali.PySurfaceResizer(vali.Y, gpu, my_stream)
        y = vali.Surface.Make(vali.Y, 1280, 720, gpu)
        assert rs.Run(y, vali.Surface.Make(vali.Y, 1000, 562, gpu))[0]    # the arena is reserved here at the latest
        assert cv.RunBatchAsync(batch, cc_ctx=cc)[0]
        shim.stream_sync(gpu, busy_stream)
        new = vali.Surface.Make(vali.Y, 1002, 564, gpu)                    # (allocated before the clock starts)
        with vali.tuning.Override(TAP_FALLBACKS=0):
            t0 = time.perf_counter()
            for _ in range(60):                                            # ~ 60 x 0.3 ms of conversions queued on the other stream
                assert cv.RunBatchAsync(batch, cc_ctx=cc)[0]
            t1 = time.perf_counter()
            assert rs.RunAsync(y, new)[0]                                  # new geometry: two tables are written, nothing is allocated
            first_call = time.perf_counter() - t1
            shim.stream_sync(gpu, busy_stream)
            busy = time.perf_counter() - t0
            shim.stream_sync(gpu, my_stream)
            assert vali.tuning.Get("TAP_FALLBACKS") == 0
        assert busy > 5e-3, busy                                           # the other stream really was busy
        assert first_call < 0.25 * busy, (first_call, busy)
    finally:
        shim.stream_sync(gpu, busy_stream)
        shim.stream_sync(gpu, my_stream)
        shim.stream_destroy(gpu, busy_stream)
        shim.stream_destroy(gpu, my_stream)


def test_two_threads_share_a_small_table_budget(vali, gpu, oracle):
    """Two host threads, each with its own stream and resizer, run different geometries concurrently against the smallest budget of tables:
    lookups, evictions and table writes interleave under the library's lock; every output is exact and nothing falls back."""
    import threading

    from vali_amd._native import shim

    rng = np.random.default_rng(8)
    sw, sh = 800, 450
    host = rng.integers(0, 256, sw * sh, dtype=np.uint8)
    src = _up(vali, gpu, vali.Y, sw, sh, host)
    errors = []

    def work(k):
        try:
            stream = shim.stream_create(gpu)
            rs = vali.PySurfaceResizer(vali.Y, gpu, stream)
            down = vali.PySurfaceDownloader(gpu, stream)
            for i in range(14):
                dw, dh = 700 - 8 * i - 2 * k, 400 - 4 * i - 2 * k
                d = vali.Surface.Make(vali.Y, dw, dh, gpu)
                assert rs.Run(src, d)[0]
                got = np.zeros(d.HostSize, np.uint8)
                assert down.Run(d, got)[0]
                assert np.array_equal(got, oracle.resize_surface(host, "Y", sw, sh, dw, dh, "lanczos")), (k, dw, dh)
            shim.stream_sync(gpu, stream)
            del rs, down
            shim.stream_destroy(gpu, stream)
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(e)))

    with vali.tuning.Override(TAP_MAX_TABLES=8, TAP_FALLBACKS=0, TAP_EVICTIONS=0):
        threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert vali.tuning.Get("TAP_FALLBACKS") == 0 and vali.tuning.Get("TAP_EVICTIONS") >= 40


def test_a_launch_never_loses_a_table_it_was_just_given(vali, gpu, oracle):
    """NV12 takes four tables per launch (two axes of two planes), planar YUV six; whatever the budget is set to (the library keeps at
    least 8), the tables of ONE launch are not evicted by that launch's own later requests -- the result stays exact."""
    from conftest import make_nv12

    sw, sh = 1280, 720
    nv = make_nv12(sw, sh, 3)
    src = _up(vali, gpu, vali.NV12, sw, sh, nv)
    rs = vali.PySurfaceResizer(vali.NV12, gpu)
    with vali.tuning.Override(TAP_MAX_TABLES=1, TAP_FALLBACKS=0):
        for k in range(10):
            dw, dh = 1000 - 6 * k, 562 - 4 * k
            d = vali.Surface.Make(vali.NV12, dw, dh, gpu)
            assert rs.Run(src, d)[0]
            assert np.array_equal(_down(vali, gpu, d), oracle.resize_surface(nv.reshape(-1), "NV12", sw, sh, dw, dh, "lanczos")), (dw, dh)
        assert vali.tuning.Get("TAP_FALLBACKS") == 0
