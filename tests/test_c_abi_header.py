"""include/vali_hip.h is a C header: a C99 translation unit that includes it and a C client that
uses it must compile with gcc (no C++ constructs, no torch / HIP types leak through)."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_header_is_valid_c99(tmp_path):
    tu = tmp_path / "tu.c"
    tu.write_text('#include "vali_hip.h"\nint main(void) { vali_surface s; (void)s; return sizeof(vali_csc) == 32 ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{ROOT / 'include'}",
                    str(tu), "-o", str(tmp_path / "tu")], check=True)
    assert subprocess.run([str(tmp_path / "tu")]).returncode == 0


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_client_compiles_and_links(tmp_path):
    """The plain-C client (tests/c_abi/abi_client.c) links against libvali_hip.so alone."""
    lib = ROOT / "vali_amd" / "libvali_hip.so"
    if not lib.exists():
        pytest.skip("library not built")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", f"-I{ROOT / 'include'}",
                    str(ROOT / "tests" / "c_abi" / "abi_client.c"), "-o", str(tmp_path / "abi_client"),
                    f"-L{lib.parent}", "-lvali_hip", f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib",
                    "-Wl,--allow-shlib-undefined"], check=True)
    assert (tmp_path / "abi_client").exists()
