"""Loader of the native layer: libvali_hip.so through the _vali_shim pybind11 module.

The product path is HIP only.  If the extension is missing this module raises at
import time -- there is no CPU fallback anywhere in vali_amd (the CPU restatement
lives in /oracle and is test infrastructure).
"""
from __future__ import annotations

import os
import sys

# PyTorch-ROCm bundles its own libamdhip64.so.7; importing torch first makes the
# process use ONE HIP runtime (same SONAME -> the loader reuses it for
# libvali_hip.so), so device pointers, streams and DLPack tensors are shared
# with torch.  torch is plumbing here (allocator interop, torch.distributed).
if "torch" not in sys.modules and not os.environ.get("VALI_NO_TORCH"):
    try:  # pragma: no cover - depends on the environment
        import torch  # noqa: F401
    except Exception:  # torch is optional for the kernels themselves
        pass

try:
    from . import _vali_shim as shim  # type: ignore
except ImportError as exc:  # pragma: no cover
    raise ImportError(
        "vali_amd: the native extension (_vali_shim / libvali_hip.so) is not built. "
        "Run `python -m vali_amd.build` (needs hipcc, targets gfx950). There is no "
        f"CPU fallback. Original error: {exc}"
    ) from exc

__all__ = ["shim"]
