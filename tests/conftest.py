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
