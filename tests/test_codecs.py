"""Decoder / encoder stand-ins (vali_amd/codecs.py): CPU-mode raw reader needs no GPU."""
import numpy as np
import pytest

from conftest import GOLDEN


def test_raw_decoder_cpu_mode(vali):
    dec = vali.PyDecoder(str(GOLDEN / "test_small_2frames.nv12"), {"video_size": "424x232"}, gpu_id=-1)
    assert (dec.Width, dec.Height, dec.Format, dec.NumFrames) == (424, 232, vali.NV12, 2)
    assert not dec.IsAccelerated and dec.HostFrameSize == 424 * 232 * 3 // 2
    raw = np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8).reshape(2, -1)
    frame = np.ndarray(shape=(0,), dtype=np.uint8)
    for i in range(2):
        ok, info = dec.DecodeSingleFrame(frame)
        assert ok and info == vali.TaskExecInfo.SUCCESS and np.array_equal(frame, raw[i])
    assert dec.DecodeSingleFrame(frame) == (False, vali.TaskExecInfo.END_OF_STREAM)
    # surface decode is refused in CPU mode (PyDecoder.cpp:98-101)
    assert dec.DecodeSingleSurface(None)[0] is False


def test_compressed_input_and_encoders_raise(vali, tmp_path):
    with pytest.raises(RuntimeError):
        vali.PyDecoder(str(tmp_path / "movie.mp4"), {}, gpu_id=-1)
    with pytest.raises(RuntimeError):
        vali.PyNvEncoder({}, 0)
    with pytest.raises(RuntimeError):
        vali.PyNvJpegEncoder(0)


@pytest.mark.gpu
def test_decode_convert_download_pipeline(vali, gpu, oracle):
    """The canonical user pipeline (samples/sample_decode_show.ipynb): decode -> NV12->RGB ->
    download, on the decoder's stream, here fed from raw NV12."""
    dec = vali.PyDecoder(str(GOLDEN / "test_small_2frames.nv12"),
                         {"f": "rawvideo", "video_size": "424x232", "pixel_format": "nv12"}, gpu_id=gpu)
    cvt = vali.PySurfaceConverter(gpu, dec.Stream)
    dwn = vali.PySurfaceDownloader(gpu, dec.Stream)
    surf = vali.Surface.Make(dec.Format, dec.Width, dec.Height, gpu)
    rgb = vali.Surface.Make(vali.RGB, dec.Width, dec.Height, gpu)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    raw = np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8).reshape(2, -1)
    n = 0
    while True:
        ok, info = dec.DecodeSingleSurface(surf)
        if not ok:
            assert info == vali.TaskExecInfo.END_OF_STREAM
            break
        assert cvt.Run(surf, rgb, cc)[0]
        out = np.zeros(rgb.HostSize, np.uint8)
        assert dwn.Run(rgb, out)[0]
        want = oracle.convert(raw[n], "NV12", "RGB", 424, 232, oracle.cvt_params(csc_variant=1))
        assert np.array_equal(out, want)
        n += 1
    assert n == 2
    assert dec.DecodeSingleFrame(np.zeros(1, np.uint8)) == (False, vali.TaskExecInfo.FAIL)


@pytest.mark.gpu
def test_frame_converter_runs_on_the_gpu(vali, gpu, oracle):
    """reference tests/test_PyFrameConverter.py:59-102 (NV12 -> RGB on ndarrays)."""
    w, h = 424, 232
    raw = np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8).reshape(2, -1)
    fc = vali.PyFrameConverter(w, h, vali.NV12, vali.RGB)
    dst = np.ndarray(shape=(0,), dtype=np.uint8)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    ok, info = fc.Run(raw[0], dst, cc)
    assert ok and dst.size == w * h * 3 and fc.Format == vali.RGB
    assert np.array_equal(dst, oracle.convert(raw[0], "NV12", "RGB", w, h, oracle.cvt_params(csc_variant=1)))
    assert fc.Run(raw[0][:-1], dst, cc) == (False, vali.TaskExecInfo.INVALID_INPUT)
