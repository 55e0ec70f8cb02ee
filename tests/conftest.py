import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def _build_if_missing():
    """A clean checkout has no native parts (the .so files are git-ignored): build them once,
    through vali_amd/build.py loaded by path (importing the package needs what is being built).
    Only MISSING products trigger this, never stale ones: a test run does not recompile."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_vali_amd_build", ROOT / "vali_amd" / "build.py")
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    if not (b.lib_path().exists() and b.shim_path().exists() and b.oracle_path().exists()):
        b.build_all(force=False)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _build_if_missing()


def make_nv12(width: int, height: int, seed: int, full_range: bool = True) -> np.ndarray:
    """Seeded synthetic NV12 frame as a (height*3/2, width) uint8 array.

    seed < 0 gives the deterministic gradient frame with full 0..255 excursions
    (saturation coverage); otherwise uniform noise (SURVEY.md 8d input recipe)."""
    rows = height + (height + 1) // 2
    if seed < 0:
        yy, xx = np.mgrid[0:rows, 0:width]
        return ((xx * 255 // max(width - 1, 1) + yy * 3) % 256).astype(np.uint8)
    rng = np.random.default_rng(1234 + seed)
    if full_range:
        return rng.integers(0, 256, (rows, width), dtype=np.uint8)
    a = np.empty((rows, width), np.uint8)
    a[:height] = rng.integers(16, 236, (height, width), dtype=np.uint8)
    a[height:] = rng.integers(16, 241, (rows - height, width), dtype=np.uint8)
    return a


def frame0_nv12():
    """Frame 0 of the reference's test.mp4 (848x464) as NV12, re-derived from the reference's own rendering of it
    (tests/golden/frame_0.jpg = NV12 -> RGB, BT.709 + MPEG, JPEG q95 4:4:4): BT.709 limited-range RGB -> YCbCr, the
    chroma sample of a 2x2 block = the mean of its four copies (nearest siting, tests/test_oracle_reference_pins.py).
    Carries the JPEG's coding noise and the frame's common offset (luma -1.43): comparisons remove a constant."""
    from PIL import Image

    rgb = np.asarray(Image.open(GOLDEN / "frame_0.jpg")).astype(np.float64)
    h, w = rgb.shape[:2]
    y = np.clip(np.rint(16 + 0.1826 * rgb[..., 0] + 0.6142 * rgb[..., 1] + 0.0620 * rgb[..., 2]), 0, 255).astype(np.uint8)
    uf = 128 - 0.1006 * rgb[..., 0] - 0.3386 * rgb[..., 1] + 0.4392 * rgb[..., 2]
    vf = 128 + 0.4392 * rgb[..., 0] - 0.3989 * rgb[..., 1] - 0.0403 * rgb[..., 2]

    def block_mean(p):
        return np.clip(np.rint(0.25 * (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2])), 0, 255).astype(np.uint8)
    nv = np.zeros((h * 3 // 2, w), np.uint8)
    nv[:h] = y
    nv[h:, 0::2], nv[h:, 1::2] = block_mean(uf), block_mean(vf)
    return nv


def psnr_offset_removed(got, gold, border=4):
    """PSNR (dB, peak 255) and mean of got - gold with `border` pixels dropped on every side and the mean removed."""
    d = (np.asarray(got, np.float64) - np.asarray(gold, np.float64))[border:-border, border:-border]
    return 10 * np.log10(255.0 ** 2 / np.mean((d - d.mean()) ** 2)), d.mean()


@pytest.fixture(scope="session")
def vali():
    import vali_amd

    return vali_amd


@pytest.fixture(scope="session")
def gpu(vali):
    """Fails (does not skip) when there is no device: -m gpu tests must really run HIP."""
    n = vali.GetNumGpus()
    assert n > 0, "no HIP device visible: GPU tests need a real MI355X"
    return 0


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.lib()
    return o
