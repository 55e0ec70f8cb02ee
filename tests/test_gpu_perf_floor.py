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


# ---- the gather kernels (resize, UD, quarter turn): same idea, same process, their own floors -------------------------
# (ratio to NV12 -> RGB at 2160p measured in this process; typical values in the comment, floor ~0.72 of them: a silent
# scratch spill, a prefetch that stopped running ahead, a fall back to a byte path all cost far more than that -- those
# were found by hand with tools/cliffs.py until round 3)
def _timed(vali, gpu, stream, fn, reps=8):
    from vali_amd._native import shim
    best = 1e9
    for _ in range(3):
        for _ in range(2):
            fn()
        e0, e1 = shim.event_create(gpu), shim.event_create(gpu)
        shim.event_record(gpu, e0, stream)
        for _ in range(reps):
            fn()
        shim.event_record(gpu, e1, stream)
        shim.event_sync(gpu, e1)
        best = min(best, shim.event_elapsed_ms(e0, e1) / reps)
        shim.event_destroy(gpu, e0)
        shim.event_destroy(gpu, e1)
    return best


def _surfaces(vali, gpu, fmt, w, h, n, fill=True):
    out = [vali.Surface.Make(fmt, w, h, gpu) for _ in range(n)]
    if fill:
        host = np.random.default_rng(1).integers(16, 236, out[0].HostSize, dtype=np.uint8)
        up = vali.PyFrameUploader(gpu)
        for s in out:
            assert up.Run(host, s)[0]
    return out


GATHER_CASES = [
    # name, typical ratio, floor, builder -> (task, batch, bytes per frame, launch)
    ("lanczos NV12 2160p->1920x1088 (2:1 along x)", 0.78, 0.55, "resize", (3840, 2160, 1920, 1088)),
    ("lanczos NV12 2160p->1936x1088 (general: specialised waves since round 5)", 0.55, 0.38, "resize", (3840, 2160, 1936, 1088)),
    ("bilinear NV12 2160p->1920x1088", 0.93, 0.66, "bilinear", (3840, 2160, 1920, 1088)),
    ("UD NV12 2160p->RGB 1080p (exact 2x)", 0.84, 0.60, "ud", (3840, 2160, 1920, 1080)),
    ("UD NV12 1080p->RGB 720p (exact 3:2)", 0.75, 0.52, "ud", (1920, 1080, 1280, 720)),
    ("UD NV12 1080p->RGB 1024x576 (any ratio)", 0.50, 0.34, "ud", (1920, 1080, 1024, 576)),
    ("rotate RGB 1080p 90 degrees", 0.75, 0.52, "rot", (1920, 1080, 1080, 1920)),
    # round 4
    ("lanczos NV12 720p->1080p (3:2 enlargement)", 0.43, 0.28, "resize", (1280, 720, 1920, 1080)),
    ("lanczos NV12 720p->1600x900 (planes that grow, register form)", 0.26, 0.15, "resize", (1280, 720, 1600, 900)),
    ("lanczos NV12 1080p->720p (3:2 both ways)", 0.77, 0.50, "resize", (1920, 1080, 1280, 720)),
    # (round 6: floor 0.47 -> 0.55.  One binary on one box moved between 1.74 and 2.21 us from process to process in the A/B of
    # profiles/r06_ab.md -- 0.79 of its best --, so 0.85 of "typical" would fail on healthy code: 0.76 x 0.79 = 0.60 is where a healthy process can land, the floor sits a tenth below)
    ("UD NV12 1080p->RGB 1080p (k_ud_lean, unchanged size)", 0.76, 0.53, "ud", (1920, 1080, 1920, 1080)),
    ("UD NV12 1918x1078->RGB 1918x1078 (ragged k_ud_lean)", 0.67, 0.42, "ud", (1918, 1078, 1918, 1078)),
    # round 5: the general Lanczos form below 2160p (0.32 / 0.18 of the roofline went unnoticed by this test)
    ("lanczos NV12 1080p->1278x718 (general, 4 slots, wide tiles)", 0.41, 0.33, "resize", (1920, 1080, 1278, 718)),   # (typical re-measured in round 6 with the r05 and the r06 library on one box: 0.40-0.43 both)
    ("lanczos NV12 1366x768->854x480 (general, small frames)", 0.30, 0.24, "resize", (1366, 768, 854, 480)),          # (0.29-0.31 with either library: the 0.42 of round 5 was a label, not a measurement of this test)
    ("rotate RGB 1080p 270 degrees (tiles anchored at the last source row since round 5; astride the sectors: 0.55)", 0.72, 0.50, "rot270",
     (1920, 1080, 1080, 1920)),
    ("lanczos RGB 720p->1080p (3:2 enlargement of packed RGB; the gather kernel it left: 0.23)", 0.58, 0.36, "resize_rgb", (1280, 720, 1920, 1080)),
    ("lanczos RGB 720p->1600x900 (packed RGB that grows, register form; the gather kernel: 0.25)", 0.39, 0.26, "resize_rgb", (1280, 720, 1600, 900)),
    # round 6
    ("rotate RGB 1080p 30 degrees (source box staged in LDS; the gather form: 0.31, misaligned DS reads: 0.50)", 0.85, 0.66, "rot30",
     (1920, 1080, 1920, 1080)),
    ("rotate RGB 1080p 10 degrees (85 % of the destination covered)", 0.66, 0.52, "rot10", (1920, 1080, 1920, 1080)),
    ("planar UD YUV420->YUV444 1080p (k_resize_up2 + luma copy; equal in the r04 / r05 A/B of profiles/r06_ab.md)", 0.94, 0.74, "udplanar",
     (1920, 1080, 1920, 1080)),
]


def test_gather_kernels_keep_their_distance_to_the_headline_kernel(vali, gpu):
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    n = 24
    cvt = vali.PySurfaceConverter(gpu)
    s0, d0 = _surfaces(vali, gpu, vali.NV12, 3840, 2160, n), _surfaces(vali, gpu, vali.RGB, 3840, 2160, n, fill=False)
    b0 = cvt.PrepareBatch(s0, d0)
    ref = 37324800 * n / (_timed(vali, gpu, cvt.Stream, lambda: cvt.RunBatchAsync(b0, cc)) * 1e-3)
    del b0, s0, d0
    slow, report = [], []
    for name, typical, floor, kind, (sw, sh, dw, dh) in GATHER_CASES:
        if kind in ("resize", "bilinear", "resize_rgb"):
            fmt = vali.RGB if kind == "resize_rgb" else vali.NV12
            task = vali.PySurfaceResizer(fmt, gpu, interpolation=vali.Interpolation.LINEAR if kind == "bilinear"
                                         else vali.Interpolation.LANCZOS)
            srcs, dsts = _surfaces(vali, gpu, fmt, sw, sh, n), _surfaces(vali, gpu, fmt, dw, dh, n, fill=False)
            nbytes = (sw * sh + dw * dh) * 3 // (1 if kind == "resize_rgb" else 2)
            b = task.PrepareBatch(srcs, dsts)
            run = lambda: task.RunBatchAsync(b)           # noqa: E731
        elif kind == "ud":
            task = vali.PySurfaceUD(gpu)
            srcs, dsts = _surfaces(vali, gpu, vali.NV12, sw, sh, n), _surfaces(vali, gpu, vali.RGB, dw, dh, n, fill=False)
            nbytes = sw * sh * 3 // 2 + dw * dh * 3
            b = task.PrepareBatch(srcs, dsts)
            run = lambda: task.RunBatchAsync(b)           # noqa: E731
        elif kind == "udplanar":
            task = vali.PySurfaceUD(gpu)
            srcs, dsts = _surfaces(vali, gpu, vali.YUV420, sw, sh, n), _surfaces(vali, gpu, vali.YUV444, dw, dh, n, fill=False)
            nbytes = sw * sh * 3 // 2 + dw * dh * 3
            b = task.PrepareBatch(srcs, dsts)
            run = lambda: task.RunBatchAsync(b)           # noqa: E731
        elif kind in ("rot30", "rot10"):
            task = vali.PySurfaceRotator(gpu)
            srcs, dsts = _surfaces(vali, gpu, vali.RGB, sw, sh, n), _surfaces(vali, gpu, vali.RGB, dw, dh, n, fill=False)
            nbytes = 2 * sw * sh * 3
            b = task.PrepareBatch(srcs, dsts)
            run = (lambda: task.RunBatchAsync(b, angle=30.0)) if kind == "rot30" else (lambda: task.RunBatchAsync(b, angle=10.0))  # noqa: E731
        else:
            task = vali.PySurfaceRotator(gpu)
            srcs, dsts = _surfaces(vali, gpu, vali.RGB, sw, sh, n), _surfaces(vali, gpu, vali.RGB, dw, dh, n, fill=False)
            nbytes = 2 * sw * sh * 3
            b = task.PrepareBatch(srcs, dsts)
            run = (lambda: task.RunBatchAsync(b, angle=270.0)) if kind == "rot270" else (lambda: task.RunBatchAsync(b, angle=90.0))  # noqa: E731
        assert run()[0]
        r = nbytes * n / (_timed(vali, gpu, task.Stream, run) * 1e-3) / ref
        report.append(f"{name}: {r:.2f} of NV12->RGB (typical {typical}, floor {floor})")
        if r < (9.0 if FLOOR > 1 else floor):
            slow.append(report[-1])
        del b, srcs, dsts
    print("\n" + "\n".join(report))
    assert not slow, f"gather kernels below their floor (NV12->RGB: {ref / 1e12:.2f} TB/s):\n" + "\n".join(slow)
