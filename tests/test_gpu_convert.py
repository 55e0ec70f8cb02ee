"""Every pair of PySurfaceConverter.Conversions() on the GPU vs the oracle: bit-exact.

Mirrors reference tests/test_PySurfaceConverter.py (:152-222 rgb_deinterleave, :306-387 p10_nv12)
and the dispatch / colour-context logic of src/TC/src/TaskConvertSurface.cpp:158-962."""
import numpy as np
import pytest

from vali_amd import tasks

pytestmark = pytest.mark.gpu

DT = {"P10": np.uint16, "P12": np.uint16}


def pairs():
    import vali_amd as vali

    return [(s.name, d.name) for s, d in vali.PySurfaceConverter.Conversions()]


def contexts(src, dst):
    """(ColorSpace, ColorRange) -> oracle params for every context the pair accepts."""
    import vali_amd as vali

    S, R = vali.ColorSpace, vali.ColorRange
    yuv_src = src in ("NV12", "YUV420", "YUV444")
    rgb_dst = dst in ("RGB", "BGR", "RGB_PLANAR")
    if src == "NV12" and rgb_dst:
        return {None: dict(csc_variant=2), (S.BT_709, R.JPEG): dict(csc_variant=2),
                (S.BT_709, R.MPEG): dict(csc_variant=1), (S.BT_601, R.JPEG): dict(csc_variant=0)}
    if src == "YUV420" and rgb_dst:
        return {None: dict(csc_variant=0), (S.BT_601, R.JPEG): dict(csc_variant=0),
                (S.BT_601, R.MPEG): dict(csc_variant=3)}
    if (src, dst) == ("YUV444", "BGR"):
        return {None: dict(csc_variant=0), (S.BT_601, R.MPEG): dict(csc_variant=3)}
    if (src, dst) == ("YUV444", "RGB"):
        return {None: dict(csc_variant=0)}
    if src in ("RGB", "BGR", "RGB_PLANAR") and dst in ("YUV444", "YUV420"):
        return {None: dict(rgb2yuv_variant=0), (S.BT_601, R.MPEG): dict(rgb2yuv_variant=1)}
    if (src, dst) == ("RGB", "Y"):
        return {None: dict(rgb2yuv_variant=0)}
    return {None: {}}


def run_pair(vali, gpu, oracle, src, dst, w, h, seed=0):
    sf, df = vali.PixelFormat[src], vali.PixelFormat[dst]
    s = vali.Surface.Make(sf, w, h, gpu)
    rng = np.random.default_rng(seed)
    if src in DT:
        host = (rng.integers(0, 1024, s.HostSize // 2, dtype=np.uint16) << 6).astype(np.uint16).view(np.uint8)
    elif src == "RGB_32F":
        host = rng.random(s.HostSize // 4, dtype=np.float32).view(np.uint8)
    else:
        host = rng.integers(0, 256, s.HostSize, dtype=np.uint8)
    assert vali.PyFrameUploader(gpu).Run(host, s)[0]
    cvt = vali.PySurfaceConverter(gpu)
    for key, okw in contexts(src, dst).items():
        cc = None if key is None else vali.ColorspaceConversionContext(*key)
        d = vali.Surface.Make(df, w, h, gpu)
        ok, info = cvt.Run(s, d, cc)
        assert ok and info == vali.TaskExecInfo.SUCCESS, (src, dst, key, info)
        got = np.zeros(d.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(d, got)[0]
        want = oracle.convert(host, src, dst, w, h, oracle.cvt_params(**okw))
        assert np.array_equal(got, want), (src, dst, key)


@pytest.mark.parametrize("pair", pairs(), ids=lambda p: f"{p[0]}-{p[1]}")
@pytest.mark.parametrize("size", [(64, 48), (848, 464), (1920, 1080)])
def test_all_pairs_aligned(vali, gpu, oracle, pair, size):
    run_pair(vali, gpu, oracle, pair[0], pair[1], *size)


@pytest.mark.parametrize("pair", pairs(), ids=lambda p: f"{p[0]}-{p[1]}")
@pytest.mark.parametrize("size", [(50, 34), (424, 232), (18, 2)])
def test_all_pairs_ragged(vali, gpu, oracle, pair, size):
    """Widths that are not a multiple of 16 take the byte-granular path of every kernel."""
    run_pair(vali, gpu, oracle, pair[0], pair[1], *size, seed=1)


def test_context_errors(vali, gpu):
    """Error codes of the colour-context switches (TaskConvertSurface.cpp:254-704)."""
    S, R, T = vali.ColorSpace, vali.ColorRange, vali.TaskExecInfo
    C = vali.ColorspaceConversionContext
    cvt = vali.PySurfaceConverter(gpu)
    mk = lambda f: vali.Surface.Make(vali.PixelFormat[f], 64, 48, gpu)
    assert cvt.Run(mk("YUV420"), mk("RGB"), C(S.BT_709, R.JPEG)) == (False, T.UNSUPPORTED_FMT_CONV_PARAMS)
    assert cvt.Run(mk("YUV444"), mk("BGR"), C(S.BT_709, R.MPEG)) == (False, T.UNSUPPORTED_FMT_CONV_PARAMS)
    assert cvt.Run(mk("YUV444"), mk("RGB"), C(S.BT_601, R.MPEG)) == (False, T.FAIL)
    assert cvt.Run(mk("YUV444"), mk("BGR"), C(S.BT_601, R.UDEF)) == (False, T.FAIL)
    assert cvt.Run(mk("RGB"), mk("YUV420"), C(S.BT_709, R.JPEG)) == (False, T.UNSUPPORTED_FMT_CONV_PARAMS)
    assert cvt.Run(mk("NV12"), mk("YUV420"), C(S.BT_601, R.UDEF)) == (False, T.UNSUPPORTED_FMT_CONV_PARAMS)
    assert cvt.Run(mk("RGB"), mk("BGR"), C(S.BT_709, R.UDEF)) == (True, T.SUCCESS)


@pytest.mark.parametrize("pair", [("RGB", "RGB_PLANAR"), ("NV12", "YUV420"), ("RGB", "YUV420"),
                                  ("YUV420", "RGB"), ("RGB", "RGB_32F")])
def test_batch_equals_single(vali, gpu, oracle, pair):
    src, dst = pair
    w, h, n = 640, 360, 5
    cvt = vali.PySurfaceConverter(gpu)
    sf, df = vali.PixelFormat[src], vali.PixelFormat[dst]
    srcs = [vali.Surface.Make(sf, w, h, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(df, w, h, gpu) for _ in range(n)]
    hosts = [np.random.default_rng(i).integers(0, 256, srcs[0].HostSize, dtype=np.uint8) for i in range(n)]
    for hst, s in zip(hosts, srcs):
        assert vali.PyFrameUploader(gpu).Run(hst, s)[0]
    assert cvt.RunBatch(srcs, dsts) == (True, vali.TaskExecInfo.SUCCESS)
    okw = list(contexts(src, dst).values())[0]
    for hst, d in zip(hosts, dsts):
        got = np.zeros(d.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(d, got)[0]
        assert np.array_equal(got, oracle.convert(hst, src, dst, w, h, oracle.cvt_params(**okw)))


def test_torch_segmentation_chain(vali, gpu, oracle):
    """reference tests/test_TorchSegmentation.py: NV12 -> RGB -> RGB_32F -> RGB_32F_PLANAR via
    RunAsync on one stream, one event wait, then torch.from_dlpack."""
    import torch

    w, h = 848, 464
    rng = np.random.default_rng(9)
    nv12 = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
    cvt = vali.PySurfaceConverter(gpu)
    fmts = [vali.NV12, vali.RGB, vali.RGB_32F, vali.RGB_32F_PLANAR]
    surfs = [vali.Surface.Make(f, w, h, gpu) for f in fmts]
    assert vali.PyFrameUploader(gpu, cvt.Stream).Run(nv12, surfs[0])[0]
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    for i in range(3):
        ok, info = cvt.RunAsync(surfs[i], surfs[i + 1], cc)
        assert ok, info
    ev = vali.CudaStreamEvent(cvt.Stream, gpu)
    ev.Record()
    ev.Wait()
    t = torch.from_dlpack(surfs[3])
    assert tuple(t.shape) == (3, h, w) and t.dtype == torch.float32
    rgb = oracle.convert(nv12, "NV12", "RGB", w, h, oracle.cvt_params(csc_variant=1))
    want = (rgb.astype(np.float32) / np.float32(255)).reshape(h, w, 3).transpose(2, 0, 1)
    assert np.array_equal(t.cpu().numpy(), want)


def _random_sizes(seed, n):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        w, h = int(rng.integers(1, 1400)) * 2, int(rng.integers(1, 300)) * 2       # even: valid for 4:2:0
        if rng.random() < 0.35:
            w = max(2, int(rng.integers(1, 6)) * 1024 + 2 * int(rng.integers(-3, 4)))  # around tile edges
        out.append((w, h))
    return out


@pytest.mark.parametrize("pair", pairs(), ids=lambda p: f"{p[0]}-{p[1]}")
@pytest.mark.parametrize("size", _random_sizes(77, 6), ids=lambda s: f"{s[0]}x{s[1]}")
def test_all_pairs_random_sizes(vali, gpu, oracle, pair, size):
    """seeded random widths / heights, a third of them a few pixels around multiples of the
    kernels' 1024-pixel wave rows: tail lanes, partial 16-pixel groups, last-row pairs."""
    run_pair(vali, gpu, oracle, pair[0], pair[1], size[0], size[1], seed=size[0] + size[1])


def test_run_async_memo_follows_surfaces_contexts_and_repointing(vali, gpu, oracle):
    """PySurfaceConverter.RunAsync keeps a one-entry memo of the last (src, dst, cc_ctx) call: alternating
    pairs, a changed colour context, a different destination format and a re-pointed (DLPack) surface must
    all produce what the un-memoised dispatch produces."""
    import torch
    from conftest import make_nv12
    w, h = 320, 180
    frames = [make_nv12(w, h, 60 + i) for i in range(3)]
    srcs = []
    for f in frames:
        s_ = vali.Surface.Make(vali.NV12, w, h, gpu)
        assert vali.PyFrameUploader(gpu).Run(f.reshape(-1), s_)[0]
        srcs.append(s_)
    cv = vali.PySurfaceConverter(gpu)
    dsts = [vali.Surface.Make(vali.RGB, w, h, gpu) for _ in range(2)]
    planar = vali.Surface.Make(vali.RGB_PLANAR, w, h, gpu)
    cc709 = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    cc601 = vali.ColorspaceConversionContext(vali.ColorSpace.BT_601, vali.ColorRange.JPEG)

    def get(surf):
        out = np.zeros(surf.HostSize, np.uint8)
        assert vali.PySurfaceDownloader(gpu).Run(surf, out)[0]
        return out

    def want(i, cc, fmt="RGB"):
        fresh = vali.Surface.Make(vali.PixelFormat[fmt], w, h, gpu)
        assert vali.PySurfaceConverter(gpu).Run(srcs[i], fresh, cc)[0]   # a new converter: no memo
        return get(fresh)

    for rep in range(3):                                     # same pair repeatedly, then alternate
        assert cv.RunAsync(srcs[0], dsts[0], cc709) == (True, vali.TaskExecInfo.SUCCESS)
    assert np.array_equal(get(dsts[0]), want(0, cc709))
    for i, d in ((1, 1), (0, 0), (2, 1), (2, 0)):
        assert cv.RunAsync(srcs[i], dsts[d], cc709)[0]
        assert np.array_equal(get(dsts[d]), want(i, cc709))
    assert cv.RunAsync(srcs[2], dsts[0], cc601)[0]           # same surfaces, other matrix
    assert np.array_equal(get(dsts[0]), want(2, cc601))
    assert cv.RunAsync(srcs[2], dsts[0])[0]                  # default context
    assert np.array_equal(get(dsts[0]), want(2, None))
    assert cv.RunAsync(srcs[2], planar)[0] and cv.RunAsync(srcs[2], planar)[0]
    assert np.array_equal(get(planar), want(2, None, "RGB_PLANAR"))
    # size mismatch after a memoised success still fails
    small = vali.Surface.Make(vali.RGB, w // 2, h // 2, gpu)
    assert cv.RunAsync(srcs[2], small) == (False, vali.TaskExecInfo.INVALID_INPUT)
    with pytest.raises(ValueError):
        cv.RunAsync(dsts[0], srcs[0])                        # RGB -> NV12 is not a reference pair
    with pytest.raises(AttributeError):
        cv.RunAsync(None, dsts[0])
