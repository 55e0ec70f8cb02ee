#!/usr/bin/env python3
"""Regenerate the reference fixtures its repository ships without (tests/data/.MISSING_LARGE_BLOBS: test.nv12,
test.rgb, test.yuv420) from tests/data/test.mp4 -- on a box that has PyAV (`import av`); the build container does
not, and nothing in the test suite depends on this script's output.

  python tools/make_reference_fixtures.py /path/to/reference/tests/data [out_dir]

  test.yuv420  the CPU decoder's frames (planar 4:2:0), all frames
  test.nv12    the same frames in the layout the GPU decoder emits (semi-planar)
  test.rgb     libswscale NV12 -> RGB24, SWS_BILINEAR, BT.709 limited range: the file reference
               tests/test_PySurfaceConverter.py:228-300 (NPP, PSNR >= 42 dB) and tests/test_PyFrameConverter.py:59-102
               (swscale, PSNR >= 44 dB) compare with
With these present, tests/test_oracle_reference_pins.py's statistical pins can be replaced by direct PSNR /
bit comparisons of the oracle against test.rgb, and the UD goldens' input frame becomes available.
"""
import sys
from pathlib import Path

import numpy as np


def main():
    try:
        import av
    except ImportError:
        raise SystemExit("PyAV (`import av`) is not installed on this box")
    src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tests/data")
    out = Path(sys.argv[2] if len(sys.argv) > 2 else ".")
    out.mkdir(parents=True, exist_ok=True)
    container = av.open(str(src / "test.mp4"))
    stream = container.streams.video[0]
    with open(out / "test.yuv420", "wb") as f420, open(out / "test.nv12", "wb") as fnv, open(out / "test.rgb", "wb") as frgb:
        n = 0
        for frame in container.decode(stream):
            f420.write(np.ascontiguousarray(frame.to_ndarray(format="yuv420p")).tobytes())
            fnv.write(np.ascontiguousarray(frame.to_ndarray(format="nv12")).tobytes())
            rgb = frame.reformat(format="rgb24", src_colorspace="ITU709", dst_colorspace="ITU709", interpolation="BILINEAR")
            frgb.write(np.ascontiguousarray(rgb.to_ndarray()).tobytes())
            n += 1
    print(f"{n} frames {stream.codec_context.width}x{stream.codec_context.height} -> {out}/test.yuv420, test.nv12, test.rgb")


if __name__ == "__main__":
    main()
