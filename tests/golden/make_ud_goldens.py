#!/usr/bin/env python3
"""Crop the reference's own UD golden files (tests/data/640x360_*.raw, produced by
reference tests/test_PySurfaceUD.py from frame 0 of test.mp4 / test_hevc10.mkv) to their
first ROWS rows and store them compressed.  Run in the build container only
(/root/reference is not on the GPU box).  These are DATA fixtures of the reference's
test-suite; the source frame they were computed from needs a video decoder and is not
reproducible offline, so the tests use them through pixel-wise identities."""
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference/tests/data")
OUT = Path(__file__).resolve().parent
W, H, ROWS = 640, 360, 120


def load(name, dtype):
    return np.fromfile(REF / f"640x360_PixelFormat.{name}.raw", dtype)


nv12 = {
    "rgb": load("NV12_PixelFormat.RGB", np.uint8).reshape(H, W, 3)[:ROWS],
    "rgb_planar": load("NV12_PixelFormat.RGB_PLANAR", np.uint8).reshape(3, H, W)[:, :ROWS],
    "rgb_32f": load("NV12_PixelFormat.RGB_32F", np.float32).reshape(H, W, 3)[:ROWS],
    "rgb_32f_planar": load("NV12_PixelFormat.RGB_32F_PLANAR", np.float32).reshape(3, H, W)[:, :ROWS],
    "yuv444": load("NV12_PixelFormat.YUV444", np.uint8).reshape(3, H, W)[:, :ROWS],
}
p10 = {
    "rgb_32f": load("P10_PixelFormat.RGB_32F", np.float32).reshape(H, W, 3)[:ROWS],
    "rgb_32f_planar": load("P10_PixelFormat.RGB_32F_PLANAR", np.float32).reshape(3, H, W)[:, :ROWS],
    "yuv444_10bit": load("P10_PixelFormat.YUV444_10bit", np.uint16).reshape(3, H, W)[:, :ROWS],
}
# UDPlanar golden (reference tests/test_PySurfaceUD.py on the CPU-decoded YUV420 frame: every plane through
# nppiResize NPPI_INTER_LANCZOS, UDSurface.cpp:33-93): the only fixture that holds NPP Lanczos output at a
# NON-integer ratio (848x464 -> 640x360 luma, 424x232 -> 640x360 chroma).  tests/test_oracle_lanczos_pin.py
planar = {"yuv444": load("YUV420_PixelFormat.YUV444", np.uint8).reshape(3, H, W)[:, :ROWS]}
np.savez_compressed(OUT / "ud_640x360_yuv420_rows120.npz", **planar)
# the same for the 10-bit source (frame 0 of test_hevc10.mkv, CPU-decoded YUV420_10bit): NPP Lanczos on 16-bit planes
# (UDSurface.cpp:60-93) -- the only reference-held output of the u16 resize path.  Its input needs a decoder (and is a
# different picture from test.mp4's: correlation 0.00 with the 8-bit golden), so offline it pins only its own value
# range; tests/test_gpu_reference_video.py compares it with the HIP path on a box that has PyAV.
planar10 = {"yuv444_10bit": load("YUV420_10bit_PixelFormat.YUV444_10bit", np.uint16).reshape(3, H, W)[:, :ROWS]}
np.savez_compressed(OUT / "ud_640x360_yuv420_10bit_rows120.npz", **planar10)
print("ud_640x360_yuv420_10bit_rows120.npz", (OUT / "ud_640x360_yuv420_10bit_rows120.npz").stat().st_size)
print("ud_640x360_yuv420_rows120.npz", (OUT / "ud_640x360_yuv420_rows120.npz").stat().st_size)
# whole-file identities, checked here once on the full files (recorded in DESIGN.md)
full_rgb = load("NV12_PixelFormat.RGB", np.uint8)
full_f = load("NV12_PixelFormat.RGB_32F", np.float32)
print("full-file RGB == clip(trunc(RGB_32F*256)):",
      np.array_equal(full_rgb, np.clip(np.trunc(full_f * 256.0), 0, 255).astype(np.uint8)))
print("test_small.yuv444 == NV12->YUV444 golden:",
      np.array_equal(np.fromfile(REF / "test_small.yuv444", np.uint8), load("NV12_PixelFormat.YUV444", np.uint8)))
np.savez_compressed(OUT / "ud_640x360_nv12_rows120.npz", **nv12)
np.savez_compressed(OUT / "ud_640x360_p10_rows120.npz", **p10)
for f in ("ud_640x360_nv12_rows120.npz", "ud_640x360_p10_rows120.npz"):
    print(f, (OUT / f).stat().st_size)
