"""Host-side API behaviour that needs no GPU: enums, format table, dispatch tables."""
import pytest

import vali_amd as vali
from vali_amd.surface import FORMATS
from vali_amd import tasks


def test_enum_values_match_reference():
    # reference: src/TC/inc/MemoryInterfaces.hpp:29-58, TC_CORE.hpp:38-52
    P = vali.PixelFormat
    assert [P.UNDEFINED, P.Y, P.RGB, P.NV12, P.YUV420, P.RGB_PLANAR, P.BGR, P.YUV444, P.RGB_32F,
            P.RGB_32F_PLANAR, P.YUV422, P.P10, P.P12, P.YUV444_10bit, P.YUV420_10bit,
            P.GRAY12] == list(range(16))
    T = vali.TaskExecInfo
    assert (T.SUCCESS, T.FAIL, T.INVALID_INPUT, T.UNSUPPORTED_FMT_CONV_PARAMS, T.NOT_SUPPORTED,
            T.SRC_DST_SIZE_MISMATCH, T.SRC_DST_FMT_MISMATCH) == (0, 1, 5, 6, 7, 9, 10)
    assert (vali.ColorSpace.BT_601, vali.ColorSpace.BT_709, vali.ColorSpace.UNSPEC) == (0, 1, 2)
    assert (vali.ColorRange.MPEG, vali.ColorRange.JPEG, vali.ColorRange.UDEF) == (0, 1, 2)


def test_export_values_like_pybind11():
    assert vali.NV12 is vali.PixelFormat.NV12
    assert vali.BT_709 is vali.ColorSpace.BT_709
    assert vali.kDLROCM == 10
    import python_vali

    assert python_vali.PySurfaceConverter is vali.PySurfaceConverter


def test_cc_ctx_defaults():
    c = vali.ColorspaceConversionContext()
    assert c.color_space == vali.ColorSpace.UNSPEC and c.color_range == vali.ColorRange.UDEF
    c = vali.ColorspaceConversionContext(vali.ColorSpace.BT_601, vali.ColorRange.MPEG)
    assert (c.color_space, c.color_range) == (0, 0)


@pytest.mark.parametrize("fmt,planes,host_size", [
    # reference: tests/test_PySurface.py:293-346 + SURVEY 2.3; W=1920,H=1080
    ("Y", 1, 1920 * 1080), ("NV12", 1, 1920 * 1620), ("P10", 1, 1920 * 1620 * 2),
    ("P12", 1, 1920 * 1620 * 2), ("YUV420", 3, 1920 * 1620), ("YUV420_10bit", 3, 1920 * 1620 * 2),
    ("YUV422", 3, 1920 * 1080 * 2), ("YUV444", 3, 1920 * 1080 * 3),
    ("YUV444_10bit", 3, 1920 * 1080 * 6), ("RGB", 1, 1920 * 1080 * 3), ("BGR", 1, 1920 * 1080 * 3),
    ("RGB_32F", 1, 1920 * 1080 * 12), ("RGB_PLANAR", 1, 1920 * 1080 * 3),
    ("RGB_32F_PLANAR", 1, 1920 * 1080 * 12)])
def test_format_table_geometry(fmt, planes, host_size):
    spec = FORMATS[vali.PixelFormat[fmt]]
    geo = spec.plane_geometry(1920, 1080)
    assert spec.num_planes == planes == len(geo)
    assert sum(w * h * spec.elem_size for w, h in geo) == host_size


def test_nv12_colour_variant_selection():
    # reference: TaskConvertSurface.cpp:117-149
    C, S, R = vali.ColorspaceConversionContext, vali.ColorSpace, vali.ColorRange
    assert tasks._nv12_variant(None) == tasks.CSC_NPP_709HDTV
    assert tasks._nv12_variant(C(S.BT_709, R.JPEG)) == tasks.CSC_NPP_709HDTV
    assert tasks._nv12_variant(C(S.BT_709, R.MPEG)) == tasks.CSC_NPP_709CSC
    assert tasks._nv12_variant(C(S.BT_709, R.UDEF)) == tasks.CSC_NPP_709CSC
    assert tasks._nv12_variant(C(S.BT_601, R.JPEG)) == tasks.CSC_NPP_YUV
    assert tasks._nv12_variant(C(S.BT_601, R.MPEG)) is None
    assert tasks._nv12_variant(C(S.UNSPEC, R.JPEG)) is None


def test_product_tables_match_oracle_tables(oracle):
    """The product's coefficient table (vali_amd/tasks.py) and the oracle's are written
    independently; they must agree to the last float bit."""
    import numpy as np

    for variant, coeffs in ((oracle.CSC_YUV, tasks.CSC_NPP_YUV),
                            (oracle.CSC_709CSC, tasks.CSC_NPP_709CSC),
                            (oracle.CSC_709HDTV, tasks.CSC_NPP_709HDTV),
                            (oracle.CSC_YCBCR, tasks.CSC_NPP_YCBCR)):
        assert np.array_equal(np.array(oracle.csc(variant).astuple(), np.float32),
                              np.array(coeffs, np.float32))
        assert tasks._csc(coeffs).astuple() == oracle.csc(variant).astuple()


def test_conversions_list_contains_reference_pairs():
    convs = vali.PySurfaceConverter.Conversions()
    assert (vali.NV12, vali.RGB) in convs and (vali.NV12, vali.BGR) in convs


def test_product_does_not_import_oracle():
    """No file under vali_amd/ may reference the oracle (CPU restatement)."""
    from pathlib import Path

    root = Path(vali.__file__).resolve().parent
    offenders = []
    for p in list(root.rglob("*.py")) + list(root.rglob("*.hip")) + list(root.rglob("*.cpp")) + \
            list(root.rglob("*.hpp")):
        if p.name == "build.py":
            continue  # builds the oracle .so, does not use it
        text = p.read_text()
        if "import oracle" in text or "from oracle" in text or "vali_oracle.h" in text:
            offenders.append(str(p))
    assert not offenders


def test_public_api_surface_of_the_hot_path():
    """Every name SURVEY.md 8(b1) lists for the path is importable from `vali_amd` and from the
    `python_vali` alias, with the reference's method names (src/python_vali/__init__.pyi)."""
    import python_vali

    api = {
        "PySurfaceConverter": ["Run", "RunAsync", "Conversions", "Stream", "RunBatch", "RunBatchAsync"],
        "PySurfaceResizer": ["Run", "RunAsync", "Stream", "RunBatch", "RunBatchAsync"],
        "PySurfaceRotator": ["Run", "RunAsync", "SupportedFormats", "Stream"],
        "PySurfaceUD": ["Run", "RunAsync", "SupportedFormats", "Stream"],
        "PySurfacePreprocessor": ["Run", "RunAsync", "RunBatch", "RunBatchAsync", "Stream"],
        "PyFrameUploader": ["Run"], "PySurfaceDownloader": ["Run"], "PyFrameConverter": ["Run", "Format"],
        "Surface": ["Make", "Clone", "Width", "Height", "Pitch", "Format", "IsEmpty", "NumPlanes", "HostSize",
                    "IsOwnMemory", "Shape", "Planes", "from_dlpack", "from_cai", "__dlpack__", "__dlpack_device__",
                    "__cuda_array_interface__"],
        "SurfacePlane": ["Width", "Height", "Pitch", "ElemSize", "HostFrameSize", "GpuMem", "__dlpack__",
                         "__dlpack_device__", "__cuda_array_interface__"],
        "CudaBuffer": ["Make", "Clone", "CopyFrom", "ElemSize", "GpuMem", "NumElems", "RawMemSize"],
        "CudaStreamEvent": ["Record", "Wait"],
    }
    for mod in (vali, python_vali):
        for cls, members in api.items():
            c = getattr(mod, cls)
            missing = [m for m in members if not hasattr(c, m)]
            assert not missing, (cls, missing)
        for name in ("ColorspaceConversionContext", "PixelFormat", "TaskExecInfo", "TaskExecDetails", "ColorSpace",
                     "ColorRange", "DLDeviceType", "GetNumGpus", "Interpolation", "NV12", "RGB", "SUCCESS", "BT_709"):
            assert hasattr(mod, name), name


def test_build_entry_points_do_not_need_the_extension_they_build():
    """A clean checkout has no .so (git-ignored): __graft_entry__.build() must load
    vali_amd/build.py by path, and `python -m vali_amd.build` must get past the package import."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(vali.__file__).resolve().parent.parent
    src = (root / "__graft_entry__.py").read_text()
    assert "spec_from_file_location" in src and "from vali_amd import build" not in src
    build_src = (root / "vali_amd" / "build.py").read_text()
    assert "from ." not in build_src and "import vali_amd" not in build_src      # stand-alone script
    # the package import is skipped only for that one command line
    code = ("import sys; sys.orig_argv = ['python', '-m', 'vali_amd.build']; import importlib, vali_amd;"
            "assert vali_amd._BUILDING and not hasattr(vali_amd, 'Surface')")
    assert subprocess.run([sys.executable, "-c", code], cwd=root).returncode == 0
    assert subprocess.run([sys.executable, "-m", "vali_amd.build"], cwd=root, capture_output=True).returncode == 0


def test_keyword_names_of_the_hot_path_match_the_reference():
    """Keyword arguments a reference user writes (src/python_vali/__init__.pyi and the py::arg
    lists of src/python_vali/src/*.cpp) are accepted under the same names."""
    import inspect

    expect = {
        (vali.Surface, "Make"): ["format", "width", "height", "gpu_id"],
        (vali.Surface, "from_dlpack"): ["capsule", "format"],
        (vali.Surface, "from_cai"): ["dict", "format"],
        (vali.PySurfaceConverter, "__init__"): ["gpu_id", "stream"],
        (vali.PySurfaceConverter, "Run"): ["src", "dst", "cc_ctx"],
        (vali.PySurfaceConverter, "RunAsync"): ["src", "dst", "cc_ctx"],
        (vali.PySurfaceResizer, "__init__"): ["format", "gpu_id", "stream"],
        (vali.PySurfaceResizer, "Run"): ["src", "dst"],
        (vali.PySurfaceRotator, "__init__"): ["gpu_id", "stream"],
        (vali.PySurfaceRotator, "Run"): ["src", "dst", "angle", "shift_x", "shift_y"],
        (vali.PySurfaceUD, "__init__"): ["gpu_id", "stream"],
        (vali.PySurfaceUD, "Run"): ["src", "dst"],
        (vali.PyFrameUploader, "__init__"): ["gpu_id", "stream"],
        (vali.PyFrameUploader, "Run"): ["src", "dst"],
        (vali.PySurfaceDownloader, "Run"): ["src", "dst"],
        (vali.PyFrameConverter, "__init__"): ["width", "height", "src_format", "dst_format"],
        (vali.PyFrameConverter, "Run"): ["src", "dst", "cc_ctx"],
        (vali.CudaStreamEvent, "__init__"): ["stream", "gpu_id"],
        (vali.CudaBuffer, "Make"): ["elem_size", "num_elems", "gpu_id"],
        (vali.ColorspaceConversionContext, "__init__"): ["color_space", "color_range"],
    }
    for (cls, name), names in expect.items():
        params = [p for p in inspect.signature(getattr(cls, name)).parameters if p != "self"]
        assert params[:len(names)] == names, (cls.__name__, name, params)


def test_resizer_default_is_the_references_filter():
    """drop-in callers write PySurfaceResizer(format, gpu_id[, stream]) and must get NPPI_INTER_LANCZOS
    (TaskResizeSurface.cpp:67); bilinear (BASELINE config 3) and bicubic are explicit opt-ins"""
    import inspect
    import vali_amd as vali

    sig = inspect.signature(vali.PySurfaceResizer.__init__)
    assert list(sig.parameters)[1:4] == ["format", "gpu_id", "stream"]
    assert sig.parameters["interpolation"].default == vali.Interpolation.LANCZOS
    assert int(vali.Interpolation.LANCZOS) == 16 and int(vali.Interpolation.LINEAR) == 1 and int(vali.Interpolation.CUBIC) == 4
