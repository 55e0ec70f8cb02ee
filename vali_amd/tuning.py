"""Tuning switches and tracing of libvali_hip.so (include/vali_hip.h: vali_tuning_key).

Every alternative kernel form kept for A/B measurements and path-coverage tests is selected through one
process-wide table.  A switch never changes a result, only which kernel produces it
(tests/test_gpu_tuning.py).  `ROCTX = 1` wraps every operator call in a roctx range named after the C entry
point -- the equivalent of the reference's NvtxMark (src/TC/inc/Tasks.hpp:32-59) -- for
`rocprofv3 --marker-trace --kernel-trace`.
"""
from __future__ import annotations

from contextlib import contextmanager

from ._native import shim

KEYS = {name[len("TUNE_"):]: getattr(shim, name) for name in dir(shim)
        if name.startswith("TUNE_") and name != "TUNE_COUNT"}


def _key(name) -> int:
    if isinstance(name, int):
        return name
    try:
        return KEYS[str(name).upper()]
    except KeyError:
        raise KeyError(f"unknown tuning switch {name!r}; known: {sorted(KEYS)}") from None


def Get(name) -> int:
    return shim.tuning_get(_key(name))


def Set(name, value: int) -> None:
    rc = shim.tuning_set(_key(name), int(value))
    if rc != 0:
        raise RuntimeError(shim.last_error())


@contextmanager
def Override(**switches):
    """with tuning.Override(UD_DOWN2=0, ROCTX=1): ...  -- restored on exit."""
    old = {k: Get(k) for k in switches}
    try:
        for k, v in switches.items():
            Set(k, v)
        yield
    finally:
        for k, v in old.items():
            Set(k, v)


def EnableTracing(on: bool = True) -> None:
    """roctx range per operator call (raises if no roctx library can be loaded)."""
    Set("ROCTX", 1 if on else 0)
