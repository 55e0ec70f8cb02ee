"""The drop-in boundary exercised from plain C (tests/c_abi/abi_client.c): no Python objects, no
torch -- allocate, upload, vali_nv12_to_rgb, vali_resize(LANCZOS), download; outputs are
compared with the oracle bit for bit."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import make_nv12

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_client_end_to_end(tmp_path, gpu, oracle):
    lib = ROOT / "vali_amd" / "libvali_hip.so"
    exe = tmp_path / "abi_client"
    subprocess.run(["gcc", "-std=c99", "-O1", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c_abi" / "abi_client.c"),
                    "-o", str(exe), f"-L{lib.parent}", "-lvali_hip", f"-Wl,-rpath,{lib.parent}",
                    "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"], check=True)
    w, h = 1280, 720
    nv = make_nv12(w, h, 77)
    (tmp_path / "in.nv12").write_bytes(nv.tobytes())
    r = subprocess.run([str(exe), str(tmp_path / "in.nv12"), str(w), str(h), str(tmp_path / "out.rgb"),
                        str(tmp_path / "half.nv12")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    assert r.stdout.startswith("ok ")
    rgb = np.fromfile(tmp_path / "out.rgb", np.uint8)
    assert np.array_equal(rgb, oracle.nv12_to_rgb(nv, w, h, oracle.csc(1), "RGB").reshape(-1))
    half = np.fromfile(tmp_path / "half.nv12", np.uint8)
    assert np.array_equal(half, oracle.resize_surface(nv.reshape(-1), "NV12", w, h, w // 2, h // 2, "lanczos"))
