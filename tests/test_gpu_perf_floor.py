"""Relative speed floor of the 8-bit converter pairs.

Parity tests cannot see a kernel that became slow.  In round 2 every pair with a packed RGB / BGR destination lost
half its speed to scratch memory for most of the round.  tests/test_kernel_resources.py guards that
cause statically; this test guards the effect: at 1080p, batch 24, one launch, every pair must move its bytes at no
less than 0.72 of the rate of NV12 -> RGB measured in the same process (the best of three short measurements each;
healthy pairs sit at 0.89-1.1, the regressed ones sat at 0.45-0.77).  Relative, so box-to-box spread cancels.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import os

FLOOR = float(os.environ.get("VALI_PERF_FLOOR", "0.72"))   # (set it to 9 to see every pair's ratio in the failure message)
W, H, N = 1920, 1080, 24


def rate(vali, gpu, cvt, up, sf, df, cc):
    from vali_amd._native import shim

    srcs = [vali.Surface.Make(sf, W, H, gpu) for _ in range(N)]
    dsts = [vali.Surface.Make(df, W, H, gpu) for _ in range(N)]
    host = np.random.default_rng(0).integers(16, 236, srcs[0].HostSize, dtype=np.uint8)
    for s in srcs:
        assert up.Run(host, s)[0]
    batch = cvt.PrepareBatch(srcs, dsts)
    stream = cvt.Stream
    best = 0.0
    for _ in range(3):
        for _ in range(3):
            cvt.RunBatchAsync(batch, cc)
        e0, e1 = shim.event_create(gpu), shim.event_create(gpu)
        shim.event_record(gpu, e0, stream)
        reps = 10
        for _ in range(reps):
            ok, info = cvt.RunBatchAsync(batch, cc)
            assert ok, info
        shim.event_record(gpu, e1, stream)
        shim.event_sync(gpu, e1)
        ms = shim.event_elapsed_ms(e0, e1) / reps
        shim.event_destroy(gpu, e0)
        shim.event_destroy(gpu, e1)
        best = max(best, (srcs[0].HostSize + dsts[0].HostSize) * N / (ms * 1e-3))
    return best


def test_every_8bit_pair_keeps_up_with_the_headline_pair(vali, gpu):
    cvt = vali.PySurfaceConverter(gpu)
    up = vali.PyFrameUploader(gpu)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    F = vali.PixelFormat
    ref = rate(vali, gpu, cvt, up, F.NV12, F.RGB, cc)
    slow = []
    for sf, df in vali.PySurfaceConverter.Conversions():
        if (sf, df) == (F.NV12, F.RGB) or F.RGB_32F in (sf, df) or F.RGB_32F_PLANAR in (sf, df) or sf in (F.P10, F.P12):
            continue    # the float / 16-bit element kernels are a different family (and 5x the bytes per frame)
        r = rate(vali, gpu, cvt, up, sf, df, cc)
        if r < FLOOR * ref:
            slow.append(f"{sf.name}->{df.name}: {r / 1e12:.2f} TB/s = {r / ref:.2f} of NV12->RGB ({ref / 1e12:.2f} TB/s)")
    assert not slow, "converter pairs below the floor:\n" + "\n".join(slow)
