"""The resize / UD kernels have a direct-gather form that normally serves only very large downscale
factors and foreign unaligned memory.  VALI_RESIZE_FORCE_GATHER=1 / VALI_UD_FORCE_GATHER=1 (read once
per process) make every geometry take it, so the ordinary parity suites can be replayed through that
code in a child process.  (This replay found a real bug: lanes without pixels left the kernel before
v_readlane broadcast the row taps of a ragged last tile from them.)"""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_parity_suites_through_the_gather_forms():
    env = dict(os.environ, VALI_RESIZE_FORCE_GATHER="1", VALI_UD_FORCE_GATHER="1")
    files = ["tests/test_gpu_resize.py", "tests/test_gpu_ud.py", "tests/test_gpu_edge_geometry.py",
             "tests/test_gpu_random_geometry.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *files],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("switch,files", [
    ("VALI_ROTATE_NO_TILE=1", ["tests/test_gpu_rotate.py"]),
    ("VALI_NV12_DIRECT_STORE=1", ["tests/test_gpu_nv12_rgb.py"]),
    ("VALI_NV12_ROWPAIRS=1", ["tests/test_gpu_nv12_rgb.py", "tests/test_gpu_convert.py"]),
    ("VALI_WAVES_PER_CU=8", ["tests/test_gpu_nv12_rgb.py", "tests/test_gpu_convert.py"]),
    ("VALI_UD_DOWN2=0", ["tests/test_gpu_ud.py", "tests/test_gpu_ud_down2.py"]),
    # widths that are not multiples of 8 leave the exact-ratio kernels by default: their ragged-lane path is replayed here
    ("VALI_UD_DOWN2=2", ["tests/test_gpu_ud.py", "tests/test_gpu_ud_down2.py", "tests/test_gpu_edge_geometry.py", "tests/test_gpu_random_geometry.py"]),
    # single surfaces and small batches take 2- or 4-row waves by themselves: the whole parity suites once more through
    # the 8-row form that batches use
    ("VALI_ROWS_PER_WAVE=8", ["tests/test_gpu_ud.py", "tests/test_gpu_ud_down2.py", "tests/test_gpu_resize.py", "tests/test_gpu_preproc.py",
                              "tests/test_gpu_edge_geometry.py", "tests/test_gpu_random_geometry.py"]),
    ("VALI_RESIZE_POINT=0", ["tests/test_gpu_resize.py"]),
    ("VALI_RESIZE_POINT=2", ["tests/test_gpu_resize.py", "tests/test_gpu_edge_geometry.py"]),
    # the Lanczos / bicubic kernel picks 2- / 8- / 32-row waves by the size of the launch: single-surface tests only
    # ever see the 2-row form, so the whole parity suites are replayed through each of the others
    ("VALI_RESIZE_NO_SEPARABLE=1", ["tests/test_gpu_resize.py", "tests/test_gpu_edge_geometry.py", "tests/test_gpu_random_geometry.py", "tests/test_gpu_ud.py"]),
    ("VALI_RESIZE_NO_SEPARABLE=3", ["tests/test_gpu_resize.py", "tests/test_gpu_edge_geometry.py", "tests/test_gpu_random_geometry.py", "tests/test_gpu_ud.py"]),
    # 8-bit planes that grow on both axes take the register form of round 4: the LDS-staged rows form once more for them
    ("VALI_RESIZE_ROWS=3", ["tests/test_gpu_resize.py", "tests/test_gpu_edge_geometry.py", "tests/test_gpu_random_geometry.py"]),
])
def test_parity_suites_under_each_ab_switch(switch, files):
    """the alternative kernel forms kept behind environment switches (tools/README.md) stay bit-exact"""
    k, v = switch.split("=")
    env = dict(os.environ, **{k: v})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "--deselect", "tests/test_gpu_ud_down2.py::test_down2_equals_the_general_kernel", *files],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
