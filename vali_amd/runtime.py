"""Per-GPU resources: one non-blocking HIP stream per device, stream events.

HIP-native counterpart of CudaResMgr / CudaStreamEvent
(reference: src/TC/inc/CudaUtils.hpp:29-135, src/TC/src/CudaUtils.cpp:35-68, 185-299).
There are no driver contexts to push on HIP: a stream carries its device.
"""
from __future__ import annotations

import threading

from ._native import shim


def GetNumGpus() -> int:
    """reference: src/python_vali/src/VALI.cpp:498 (CudaResMgr::GetNumGpus)."""
    return shim.device_count()


class HipResMgr:
    """Lazy singleton: stream per GPU, created on first use (CudaUtils.cpp:222-238)."""

    _instance = None
    _lock = threading.Lock()

    def __init__(self):
        self._streams: dict[int, int] = {}
        self._mtx = threading.Lock()

    @classmethod
    def Instance(cls) -> "HipResMgr":
        with cls._lock:
            if cls._instance is None:
                cls._instance = cls()
            return cls._instance

    def GetNumGpus(self) -> int:
        return shim.device_count()

    def _check(self, gpu_id: int) -> None:
        n = shim.device_count()
        if gpu_id < 0 or gpu_id >= n:
            raise RuntimeError(f"GPU id {gpu_id} is out of range (found {n} HIP device(s))")

    def GetStream(self, gpu_id: int) -> int:
        with self._mtx:
            s = self._streams.get(gpu_id)
            if s is None:
                self._check(gpu_id)
                s = shim.stream_create(gpu_id)
                self._streams[gpu_id] = s
            return s

    def GetCtx(self, gpu_id: int) -> int:
        """HIP has no CUcontext; the 'context' of the API is the device ordinal."""
        self._check(gpu_id)
        return gpu_id


class CudaStreamEvent:
    """Stream event: Record() then Wait() blocks the host until the stream reaches it.

    reference: src/TC/src/CudaUtils.cpp:35-68, binding src/python_vali/src/VALI.cpp:281-315.
    Name kept for API compatibility; the object is a hipEvent_t.
    """

    def __init__(self, stream: int, gpu_id: int):
        self._gpu_id = int(gpu_id)
        self._stream = int(stream)
        self._event = shim.event_create(self._gpu_id)

    def Record(self) -> None:
        shim.event_record(self._gpu_id, self._event, self._stream)

    def Wait(self) -> None:
        shim.event_sync(self._gpu_id, self._event)

    @property
    def Handle(self) -> int:
        return self._event

    def __del__(self):
        ev = getattr(self, "_event", 0)
        if ev:
            try:
                shim.event_destroy(self._gpu_id, ev)
            except Exception:
                pass
            self._event = 0


_capturing = set()      # (gpu_id, stream) pairs between StreamCapture.__enter__ and __exit__


def is_capturing(gpu_id: int, stream: int) -> bool:
    return (int(gpu_id), int(stream)) in _capturing


class StreamCapture:
    """Record a chain of asynchronous task calls on one stream, replay it as ONE launch.

    New (no reference counterpart): per-frame chains of small launches -- BASELINE config 2,
    1080p at batch 1 -- are bound by the ~5 us host cost of every launch.  Inside the `with`
    block every `RunAsync` / `RunBatchAsync` issued on `stream` is recorded into a hipGraph
    (surfaces are frozen by address: replay always works on the same Surface objects, which
    the caller refills between launches, e.g. by decoding into the same surface);
    `Launch()` then submits the whole chain at once.

        cap = vali.StreamCapture(stream, gpu_id)
        with cap:
            resizer.RunAsync(frame, small)
            converter.RunAsync(small, rgb, cc)
        for _ in frames:
            ...refill `frame`...
            cap.Launch()

    Only asynchronous calls may appear in the block (`Run` synchronises, which is illegal
    while capturing); keep the captured surfaces alive as long as the capture.
    """

    def __init__(self, stream: int, gpu_id: int):
        self._stream, self._gpu_id = int(stream), int(gpu_id)
        if not self._stream:
            raise ValueError("StreamCapture needs an explicit stream")
        self._graph = 0
        self._keep = []

    def Keep(self, *objects) -> "StreamCapture":
        """Tie the lifetime of surfaces / batches to the capture."""
        self._keep.extend(objects)
        return self

    def __enter__(self) -> "StreamCapture":
        if self._graph:
            raise RuntimeError("StreamCapture: already captured")
        shim.graph_capture_begin(self._gpu_id, self._stream)
        _capturing.add((self._gpu_id, self._stream))
        return self

    def __exit__(self, exc_type, exc, tb) -> bool:
        _capturing.discard((self._gpu_id, self._stream))
        try:
            g = shim.graph_capture_end(self._gpu_id, self._stream)
        except Exception:
            if exc_type is None:
                raise
            return False
        if exc_type is None:
            self._graph = g
        else:
            shim.graph_destroy(self._gpu_id, g)
        return False

    def Launch(self) -> None:
        if not self._graph:
            raise RuntimeError("StreamCapture: nothing captured")
        shim.graph_launch(self._gpu_id, self._graph, self._stream)

    def __del__(self):
        if getattr(self, "_graph", 0):
            try:
                shim.graph_destroy(self._gpu_id, self._graph)
            except Exception:
                pass
            self._graph = 0
