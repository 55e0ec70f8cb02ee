"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    names = []
    for hdr in sorted((ROOT / "include").glob("*.h")):
        text = re.sub(r"/\*.*?\*/", "", hdr.read_text(), flags=re.S)
        names += re.findall(r"VALI_API\s+[\w\s\*]+?\b(vali_\w+)\s*\(", text)
    return names


def test_header_declares_something():
    syms = declared_symbols()
    assert len(syms) >= 20
    assert "vali_nv12_to_rgb" in syms and "vali_nv12_to_rgb_batch" in syms


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(str(ROOT / "vali_amd" / "libvali_hip.so"))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/ but not exported: {missing}"


def test_library_reports_version_and_no_device_cleanly():
    lib = ctypes.CDLL(str(ROOT / "vali_amd" / "libvali_hip.so"))
    lib.vali_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.vali_version()
    n = ctypes.c_int(-1)
    rc = lib.vali_device_count(ctypes.byref(n))
    assert rc in (0, -4) and n.value >= 0


def test_invalid_arguments_are_rejected_without_a_gpu():
    """Argument validation happens before any HIP call."""
    lib = ctypes.CDLL(str(ROOT / "vali_amd" / "libvali_hip.so"))
    lib.vali_last_error.restype = ctypes.c_char_p
    assert lib.vali_nv12_to_rgb(None, None, None, None) == -1
    assert b"null" in lib.vali_last_error()
    assert lib.vali_mem_alloc_pitch(0, 0, 0, None, None) == -1


def test_shim_struct_size_matches_header():
    from vali_amd._native import shim

    # 3 pointers + 3 pitches + width + height + format, padded to 8
    assert shim.SURFACE_DESC_SIZE == 48


def test_tuning_table_round_trip_without_a_gpu():
    """vali_tuning_set / vali_tuning_get work without a device; unknown keys are refused; defaults as documented."""
    from vali_amd._native import shim

    assert shim.TUNE_COUNT == 19
    defaults = {shim.TUNE_RESIZE_POINT: 1, shim.TUNE_UD_DOWN2: 1, shim.TUNE_UD_OCC5: 0, shim.TUNE_RESIZE_ROWS: 1,
                shim.TUNE_TAP_MAX_TABLES: 1024}
    import os
    if not any(k.startswith("VALI_") and k not in ("VALI_BENCH_BACKEND", "VALI_NO_TORCH") for k in os.environ):
        for k in range(shim.TUNE_COUNT):
            assert shim.tuning_get(k) == defaults.get(k, 0), k
    old = shim.tuning_get(shim.TUNE_WAVES_PER_CU)
    assert shim.tuning_set(shim.TUNE_WAVES_PER_CU, 24) == 0 and shim.tuning_get(shim.TUNE_WAVES_PER_CU) == 24
    assert shim.tuning_set(shim.TUNE_WAVES_PER_CU, old) == 0
    assert shim.tuning_set(shim.TUNE_COUNT, 1) == shim.ERR_INVALID_ARG
    assert shim.tuning_set(-1, 1) == shim.ERR_INVALID_ARG
    import vali_amd
    with vali_amd.tuning.Override(UD_DOWN2=0):
        assert vali_amd.tuning.Get("ud_down2") == 0
    assert vali_amd.tuning.Get("UD_DOWN2") == 1
    import pytest
    with pytest.raises(KeyError):
        vali_amd.tuning.Get("NO_SUCH_SWITCH")
