"""vali_amd -- MI355X-native surface processing behind the python_vali API.

`import vali_amd as vali` (or `import python_vali as vali`) gives the reference's
names: Surface, SurfacePlane, PixelFormat, PySurfaceConverter, PyFrameUploader, ...
(reference: src/python_vali/__init__.py:14-16 re-exporting _python_vali).
"""
import sys as _sys

# `python -m vali_amd.build` has to import this package before it can build the extension the
# package needs: in that one situation (and only then) the imports below are skipped.
_BUILDING = "vali_amd.build" in getattr(_sys, "orig_argv", [])
if not _BUILDING:
    from ._native import shim as _shim  # noqa: F401  (fails loudly if the HIP library is absent)
    from .codecs import (NvJpegEncodeContext, PacketData, PyDecoder, PyFrameConverter, PyNvEncoder, PyNvJpegEncoder,
                         SeekContext, SetFFMpegLogLevel, StreamParams)
    from .enums import (ColorRange, ColorSpace, ColorspaceConversionContext, DecodeMode, DLDeviceType, FfmpegLogLevel,
                        Interpolation, PixelFormat, TaskExecDetails, TaskExecInfo, TaskExecStatus, export_values)
    from .runtime import CudaStreamEvent, GetNumGpus, HipResMgr, StreamCapture
    from .surface import Surface, SurfacePlane
    from .buffer import CudaBuffer
    from .tasks import (PySurfaceConverter, PySurfacePreprocessor, PySurfaceResizer, PySurfaceRotator, PySurfaceUD,
                        SurfaceBatch)
    from .pipeline import BatchedFramePipeline, broadcast_coefficients, shard_frames
    from .transfer import PyFrameUploader, PySurfaceDownloader
    from . import tuning

    export_values(globals())

    __version__ = "0.1.0"
