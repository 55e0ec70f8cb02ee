"""Batched, frame-sharded surface pipeline: one process per GPU.

New design (the reference has no multi-GPU code: SURVEY.md 2.1 / 8e).  Frames are
independent, so N frames are split into contiguous blocks, one per rank; no frame ever
crosses xGMI.  The only exchange step is a broadcast of the 32-byte colour-coefficient
block (vali_csc) from rank 0 over torch.distributed (backend "nccl" = RCCL on ROCm;
"gloo" in the CPU tests), so that every GPU converts with the same matrix.

Two ways to feed a rank's GPU:
* resident: the shard's surfaces are filled once and converted with one launch per step (`run_async`) -- what the
  headline number measures (inputs already in HBM);
* host-fed: `IngestRing` -- K slots of pinned host memory + device surfaces; the H2D copy of slot i+1 runs on a copy
  stream while slot i is converted on the task's stream (events order the two; no host synchronisation per slot).  This
  is what a decoder-fed pipeline delivers: bound by PCIe (53-55 GB/s / 12.4 MB = ~4.3 k 2160p NV12 frames/s per GPU).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import tasks
from .enums import ColorspaceConversionContext, Interpolation, PixelFormat, TaskExecInfo
from .surface import Surface


def shard_frames(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of the contiguous block of frames rank `rank` owns
    (frame i -> rank i // ceil(total/world))."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    per = -(-total // world)
    begin = min(rank * per, total)
    return begin, min(begin + per, total)


def broadcast_coefficients(coeffs: Optional[Sequence[float]], src: int = 0, device=None,
                           group=None) -> Tuple[float, ...]:
    """Broadcast the 6 colour coefficients (padded to the 8-float vali_csc block) from
    rank `src`.  Without an initialised process group this is the identity."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        if coeffs is None:
            raise ValueError("no coefficients and no process group to receive them from")
        return tuple(float(np.float32(c)) for c in coeffs)
    block = torch.zeros(8, dtype=torch.float32, device=device if device is not None else "cpu")
    if dist.get_rank(group) == src:
        if coeffs is None:
            raise ValueError("source rank must provide the coefficients")
        block[:6] = torch.tensor([float(c) for c in coeffs], dtype=torch.float32)
    dist.broadcast(block, src=src, group=group)
    return tuple(float(x) for x in block[:6].cpu().numpy())


# ---- NUMA placement of a rank's host thread ---------------------------------------------------------------------
def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_cpus(pci_bus_id: str, sysfs: str = "/sys") -> Optional[List[int]]:
    """CPUs of the NUMA node the GPU at PCI address `pci_bus_id` ("0000:c1:00.0") hangs off, from sysfs
    (/sys/bus/pci/devices/<id>/numa_node, /sys/devices/system/node/node<N>/cpulist); None when the platform does not say
    (single-node hosts report -1)."""
    try:
        dev = pci_bus_id.strip().lower()
        if dev.count(":") == 1:
            dev = "0000:" + dev
        with open(os.path.join(sysfs, "bus/pci/devices", dev, "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        return cpus or None
    except (OSError, ValueError):
        return None


def bind_to_gpu_numa(gpu_id: int, sysfs: str = "/sys", pci_bus_id: Optional[str] = None) -> Optional[List[int]]:
    """Restrict the calling process to the CPUs next to `gpu_id` (its uploads then come from local memory and its launch
    thread does not hop sockets): one rank per GPU calls this once.  Returns the CPU list it bound to, None when the
    topology is unknown or the platform has no affinity call -- never an error."""
    if pci_bus_id is None:
        from ._native import shim
        try:
            pci_bus_id = shim.device_pci_bus_id(int(gpu_id))
        except Exception:
            return None
    cpus = gpu_numa_cpus(pci_bus_id, sysfs)
    if not cpus or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if not allowed:
        return None
    os.sched_setaffinity(0, allowed)
    return allowed


# ---- clock / power state of a GPU (telemetry for the bench line) -------------------------------------------------
def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _dpm_current_mhz(text: Optional[str]) -> Optional[int]:
    """The starred level of an amdgpu pp_dpm_* file ("0: 132Mhz\n1: 2100Mhz *"), in MHz."""
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            digits = "".join(ch for ch in line.split(":", 1)[-1] if ch.isdigit())
            return int(digits) if digits else None
    return None


def gpu_clock_state(pci_bus_id: str, sysfs: str = "/sys") -> dict:
    """What amdgpu's sysfs says about the GPU at `pci_bus_id` RIGHT NOW: shader / memory / fabric clock (MHz), the
    power cap and the present draw (W), the hottest sensor (deg C), the compute / memory partition modes.  Every field is
    None when the platform does not expose it -- never an error: this is a note next to a measurement (the same binary
    measured 0.76 - 0.82 of the HBM peak across the boxes of the pool; the bench line carries the box's state so that such
    a spread has a cause next to it)."""
    dev = pci_bus_id.strip().lower()
    if dev.count(":") == 1:
        dev = "0000:" + dev
    base = os.path.join(sysfs, "bus/pci/devices", dev)
    out = {"sclk_mhz": _dpm_current_mhz(_read(os.path.join(base, "pp_dpm_sclk"))),
           "mclk_mhz": _dpm_current_mhz(_read(os.path.join(base, "pp_dpm_mclk"))),
           "fclk_mhz": _dpm_current_mhz(_read(os.path.join(base, "pp_dpm_fclk"))),
           "power_cap_w": None, "power_w": None, "temp_c": None,
           "compute_partition": _read(os.path.join(base, "current_compute_partition")),
           "memory_partition": _read(os.path.join(base, "current_memory_partition")),
           "perf_level": _read(os.path.join(base, "power_dpm_force_performance_level"))}
    try:
        hw = os.path.join(base, "hwmon")
        for name in sorted(os.listdir(hw)):
            h = os.path.join(hw, name)
            cap = _read(os.path.join(h, "power1_cap"))
            draw = _read(os.path.join(h, "power1_average")) or _read(os.path.join(h, "power1_input"))
            temps = [int(t) for t in (_read(os.path.join(h, f"temp{i}_input")) for i in range(1, 9)) if t and t.lstrip("-").isdigit()]
            if cap and cap.isdigit():
                out["power_cap_w"] = round(int(cap) / 1e6, 1)
                dflt = _read(os.path.join(h, "power1_cap_default"))
                if dflt and dflt.isdigit():
                    out["power_cap_default_w"] = round(int(dflt) / 1e6, 1)
            if draw and draw.isdigit():
                out["power_w"] = round(int(draw) / 1e6, 1)
            if temps:
                out["temp_c"] = round(max(temps) / 1e3, 1)
            if out["sclk_mhz"] is None:
                f1 = _read(os.path.join(h, "freq1_input"))
                if f1 and f1.isdigit():
                    out["sclk_mhz"] = int(f1) // 1000000
            if out["mclk_mhz"] is None:
                f2 = _read(os.path.join(h, "freq2_input"))
                if f2 and f2.isdigit():
                    out["mclk_mhz"] = int(f2) // 1000000
    except OSError:
        pass
    return out


def sample_clocks_under_load(pci_bus_id: str, step: Callable[[], None], seconds: float = 0.25, sysfs: str = "/sys") -> dict:
    """Repeat `step` (one launch + its completion) for `seconds` and read the GPU's state between launches: min / max of
    the clocks, the highest power draw and temperature seen -- the state the chip HOLDS under this load, taken right
    after (never inside) a timed region."""
    import time

    idle = gpu_clock_state(pci_bus_id, sysfs)
    seen: List[dict] = []
    t0 = time.perf_counter()
    while True:
        step()
        seen.append(gpu_clock_state(pci_bus_id, sysfs))
        if time.perf_counter() - t0 >= seconds:
            break

    def span(key):
        v = [s[key] for s in seen if s.get(key) is not None]
        return ([min(v), max(v)] if v else None)
    return {"under_load": {"sclk_mhz": span("sclk_mhz"), "mclk_mhz": span("mclk_mhz"), "fclk_mhz": span("fclk_mhz"),
                           "power_w": span("power_w"), "temp_c": span("temp_c"), "samples": len(seen)},
            "before": {k: idle[k] for k in ("sclk_mhz", "mclk_mhz", "power_w", "temp_c")},
            "power_cap_w": idle["power_cap_w"], "power_cap_default_w": idle.get("power_cap_default_w"), "perf_level": idle["perf_level"],
            "compute_partition": idle["compute_partition"], "memory_partition": idle["memory_partition"]}


# ---- the operators a pipeline can run -------------------------------------------------------------------------
OPS = ("convert", "resize", "ud", "preproc")


def op_geometry(op: str, width: int, height: int, dst_format: PixelFormat, dst_size=None):
    """(destination format, (dst_w, dst_h), algorithmic bytes per frame) of `op` on NV12 width x height frames."""
    if op not in OPS:
        raise ValueError(f"op must be one of {OPS}")
    dw, dh = (width, height) if op == "convert" or dst_size is None else dst_size
    src = width * height * 3 // 2
    if op == "convert":
        fmt = dst_format
        out = width * height * 3
    elif op == "resize":
        fmt, out = PixelFormat.NV12, dw * dh * 3 // 2
    elif op == "ud":
        fmt, out = PixelFormat.RGB, dw * dh * 3
    else:
        fmt, out = PixelFormat.RGB_32F_PLANAR, dw * dh * 12
    return fmt, (int(dw), int(dh)), src + out


class BatchedFramePipeline:
    """Owns this rank's shard of surfaces and runs `op` over it with one launch per step.

    op = "convert"  NV12 -> dst_format (RGB / BGR / RGB_PLANAR) at the same size: PySurfaceConverter, the headline
         "resize"   NV12 -> NV12 at dst_size, the reference's filter (Lanczos): PySurfaceResizer
         "ud"       NV12 -> RGB at dst_size: PySurfaceUD
         "preproc"  NV12 -> normalised RGB_32F_PLANAR at dst_size: PySurfacePreprocessor
    """

    def __init__(self, gpu_id: int, width: int, height: int, frames: int,
                 dst_format: PixelFormat = PixelFormat.RGB, stream=None, op: str = "convert", dst_size=None):
        self.gpu_id, self.width, self.height, self.frames, self.op = gpu_id, width, height, frames, op
        self.dst_format, self.dst_size, self.bytes_per_frame = op_geometry(op, width, height, dst_format, dst_size)
        if op == "convert":
            self.task = tasks.PySurfaceConverter(gpu_id, stream)
        elif op == "resize":
            self.task = tasks.PySurfaceResizer(PixelFormat.NV12, gpu_id, stream, interpolation=Interpolation.LANCZOS)
        elif op == "ud":
            self.task = tasks.PySurfaceUD(gpu_id, stream)
        else:
            self.task = tasks.PySurfacePreprocessor(gpu_id, stream, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), div=255.0)
        self.converter = self.task            # (the name round 1-2 callers use)
        self.srcs, self.dsts, self.batch = self._make_batch(frames)
        self._csc = None
        self._cc = None

    def _make_batch(self, n: int):
        srcs = [Surface.Make(PixelFormat.NV12, self.width, self.height, self.gpu_id) for _ in range(n)]
        dsts = [Surface.Make(self.dst_format, self.dst_size[0], self.dst_size[1], self.gpu_id) for _ in range(n)]
        return srcs, dsts, self.task.PrepareBatch(srcs, dsts)

    @property
    def Stream(self) -> int:
        return self.task.Stream

    def set_coefficients(self, cc_ctx: Optional[ColorspaceConversionContext], src: int = 0,
                         device=None, group=None) -> Tuple[float, ...]:
        """Rank `src` resolves cc_ctx to a matrix (same switch as nv12_rgb,
        TaskConvertSurface.cpp:117-149); everybody receives it."""
        import torch.distributed as dist

        is_src = not (dist.is_available() and dist.is_initialized()) or dist.get_rank(group) == src
        coeffs = tasks._nv12_variant(cc_ctx) if is_src else None
        if is_src and coeffs is None:
            raise ValueError("unsupported colour conversion parameters")
        coeffs = broadcast_coefficients(coeffs, src, device, group)
        self._csc = tasks._csc(coeffs)
        self._cc = cc_ctx if cc_ctx is not None else ColorspaceConversionContext()
        return coeffs

    def _launch(self, batch) -> Tuple[bool, TaskExecInfo]:
        if self.op == "convert":
            if self._csc is None:
                raise RuntimeError("set_coefficients() first")
            return self.task.RunBatchAsync(batch, csc=self._csc)
        if self.op == "preproc":
            return self.task.RunBatchAsync(batch, cc_ctx=self._cc)
        return self.task.RunBatchAsync(batch)

    def run_async(self) -> Tuple[bool, TaskExecInfo]:
        return self._launch(self.batch)

    def run(self) -> Tuple[bool, TaskExecInfo]:
        r = self.run_async()
        self.task._sync()
        return r

    def ingest_ring(self, slots: int = 3, frames_per_slot: Optional[int] = None, backend=None) -> "IngestRing":
        return IngestRing(self, slots, frames_per_slot or min(self.frames, 16), backend)


# ---- host-fed operation: pinned staging + overlapped H2D ---------------------------------------------------------
class _HipBackend:
    """the device side of the ring: pinned host memory, a copy stream, events (vali_host_alloc, vali_stream_wait_event ...)"""

    def __init__(self, gpu_id: int):
        from ._native import shim
        self.shim, self.gpu = shim, int(gpu_id)
        self.copy_stream = shim.stream_create(self.gpu)
        self._host = []

    def host_buffer(self, nbytes: int) -> np.ndarray:
        import ctypes
        p = self.shim.host_alloc(self.gpu, int(nbytes))
        self._host.append(p)
        return np.ctypeslib.as_array((ctypes.c_uint8 * int(nbytes)).from_address(p))

    def new_event(self):
        return self.shim.event_create(self.gpu)

    def free_event(self, event) -> None:
        self.shim.event_destroy(self.gpu, event)

    def upload(self, host: np.ndarray, frame_bytes: int, srcs: Sequence[Surface]) -> None:
        base = host.ctypes.data
        for i, s in enumerate(srcs):
            off = 0
            for p in s._planes:
                row = p.Width * p.ElemSize
                self.shim.memcpy2d_async(self.gpu, p.GpuMem, p.Pitch, base + i * frame_bytes + off, row, row, p.Height, 0,
                                         self.copy_stream)
                off += row * p.Height

    def record(self, event, stream) -> None:
        self.shim.event_record(self.gpu, event, stream)

    def stream_wait(self, stream, event) -> None:
        self.shim.stream_wait_event(self.gpu, stream, event)

    def host_wait(self, event) -> None:
        self.shim.event_sync(self.gpu, event)

    def is_done(self, event) -> bool:
        return bool(self.shim.event_query(self.gpu, event))

    def close(self) -> None:
        for p in self._host:
            try:
                self.shim.host_free(self.gpu, p)
            except Exception:
                pass
        self._host = []
        if self.copy_stream:
            try:
                self.shim.stream_destroy(self.gpu, self.copy_stream)
            except Exception:
                pass
            self.copy_stream = 0


class _Slot:
    __slots__ = ("host", "srcs", "dsts", "batch", "uploaded", "done", "busy", "tag", "valid")


class IngestRing:
    """Host frames -> GPU operator, K slots deep.

        ring = pipe.ingest_ring(slots=3, frames_per_slot=16)
        for chunk_id, chunk in enumerate(decoder):         # chunk: up to 16 frames
            for s in ring.reap():                          # finished slots, oldest first; blocks only when the slot that is
                consume(s.dsts, s.tag)                     #   about to be reused has not finished yet
            slot = ring.acquire()
            slot.host[:n * ring.frame_bytes] = ...         # the decoder writes straight into pinned memory
            ring.submit(slot, tag=chunk_id)                # H2D on the copy stream, then the operator on the task's stream
        for s in ring.drain():
            consume(s.dsts, s.tag)

    (`feed(chunks)` is that loop as a generator.)  Order per slot: [copy stream] H2D of the slot -> event `uploaded`;
    [task stream] waits `uploaded`, runs the operator on the slot's batch -> event `done`.  The copy stream never waits for
    the task stream, so the upload of slot i+1 overlaps the conversion of slot i.  A slot's outputs must be consumed before
    the slot is acquired again: reap() hands them out first, acquire() refuses a slot whose outputs were not reaped.
    `backend` abstracts the device calls (tests run the ring's ordering logic without a GPU)."""

    def __init__(self, pipe: BatchedFramePipeline, slots: int = 3, frames_per_slot: int = 16, backend=None):
        if slots < 2:
            raise ValueError("an ingest ring needs at least 2 slots to overlap anything")
        self.pipe, self.n = pipe, int(frames_per_slot)
        self.frame_bytes = pipe.width * pipe.height * 3 // 2
        self.backend = backend if backend is not None else _HipBackend(pipe.gpu_id)
        self.slots: List[_Slot] = []
        for _ in range(slots):
            s = _Slot()
            s.host = self.backend.host_buffer(self.n * self.frame_bytes)
            s.srcs, s.dsts, s.batch = pipe._make_batch(self.n)
            s.uploaded, s.done = self.backend.new_event(), self.backend.new_event()
            s.busy, s.tag, s.valid = False, None, 0
            self.slots.append(s)
        self._next = 0
        self._closed = False
        self._pending: List[_Slot] = []        # submitted, outputs not handed out yet; submission order
        self.frames_submitted = 0

    def reap(self, make_room: bool = True) -> List[_Slot]:
        """slots whose outputs are ready, oldest first.  make_room: if the slot acquire() would hand out next is still
        pending, wait for it (and everything submitted before it) -- the only place the host blocks."""
        out: List[_Slot] = []
        must = self.slots[self._next] if make_room else None
        while self._pending:
            s = self._pending[0]
            blocking = must is not None and must in self._pending
            if not blocking and not self.backend.is_done(s.done):
                break
            if blocking:
                self.backend.host_wait(s.done)
            self._pending.pop(0)
            s.busy = False
            out.append(s)
        return out

    def acquire(self) -> _Slot:
        if self._closed:
            raise RuntimeError("IngestRing is closed")
        s = self.slots[self._next]
        if s in self._pending:
            raise RuntimeError("IngestRing.acquire: the next slot still holds outputs nobody took -- call reap() first")
        self._next = (self._next + 1) % len(self.slots)
        return s

    def submit(self, slot: _Slot, tag=None, valid: Optional[int] = None) -> Tuple[bool, TaskExecInfo]:
        """`valid`: how many of the slot's n frames the producer really wrote (a final chunk may be short); the operator
        still runs on the whole slot (one prepared batch), `slot.valid` tells the consumer how many outputs count."""
        if self._closed:
            raise RuntimeError("IngestRing is closed")
        slot.valid = self.n if valid is None else max(0, min(int(valid), self.n))
        b = self.backend
        b.upload(slot.host, self.frame_bytes, slot.srcs)
        b.record(slot.uploaded, b.copy_stream)
        b.stream_wait(self.pipe.Stream, slot.uploaded)
        r = self.pipe._launch(slot.batch)
        b.record(slot.done, self.pipe.Stream)
        slot.busy, slot.tag = True, tag
        self._pending.append(slot)
        self.frames_submitted += self.n
        return r

    def drain(self) -> List[_Slot]:
        out = []
        while self._pending:
            s = self._pending.pop(0)
            self.backend.host_wait(s.done)
            s.busy = False
            out.append(s)
        return out

    def feed(self, chunks, fill: Optional[Callable] = None):
        """The loop of the class comment as a generator: every item of `chunks` is a uint8 array of 1..n whole frames (or
        anything `fill(slot.host, item)` understands; `fill` returns None -- it filled all n frames of the slot -- or the
        NUMBER of frames it wrote, an int in 1..n: a bool or anything else is a TypeError, not a count); yields
        (tag, dsts) of finished slots in submission order -- only the VALID outputs: a final chunk shorter than a slot
        yields that many surfaces, never the stale frames an earlier chunk left in the slot's tail.  The consumer runs
        between two submissions, i.e. before the slot it is looking at can be reused."""
        for tag, item in enumerate(chunks):
            for s in self.reap():
                yield s.tag, s.dsts[:s.valid]
            # A chunk the ring cannot take: what was submitted is still delivered, in order, before the error surfaces.
            err = None
            if fill is None:                       # validated BEFORE a slot is taken
                a = np.asarray(item, np.uint8).reshape(-1)
                if a.size == 0:
                    break                          # an empty chunk is the end of the input
                if a.size % self.frame_bytes or a.size > self.n * self.frame_bytes:
                    err = ValueError(f"IngestRing.feed: a chunk is 1..{self.n} whole frames of {self.frame_bytes} bytes")
            if err is None:
                slot = self.acquire()
                if fill is not None:
                    wrote = fill(slot.host, item)
                    if wrote is None:
                        valid = self.n
                    elif isinstance(wrote, (int, np.integer)) and not isinstance(wrote, (bool, np.bool_)) and 0 <= int(wrote) <= self.n:
                        valid = int(wrote)
                        if valid == 0:             # fill() wrote nothing: end of the stream (the slot goes back unused)
                            self._next = (self._next - 1) % len(self.slots)
                            break
                    else:
                        self._next = (self._next - 1) % len(self.slots)
                        err = TypeError(f"IngestRing.feed: fill() returns None or the number of frames it wrote (0 = end of input, "
                                        f"1..{self.n}), got {wrote!r}")
                else:
                    slot.host[:a.size] = a
                    valid = a.size // self.frame_bytes
            if err is None:
                ok, info = self.submit(slot, tag, valid)
                if not ok:
                    err = RuntimeError(f"IngestRing: operator failed: {info}")
            if err is not None:
                for s in self.drain():
                    yield s.tag, s.dsts[:s.valid]
                raise err
        for s in self.drain():
            yield s.tag, s.dsts[:s.valid]

    def close(self) -> None:
        """Waits for everything submitted, then releases the pinned buffers and the events.  The slots' numpy views point
        INTO the pinned allocations, so they are dropped first: after close() `slot.host` is None and acquire / submit
        raise -- a late write can no longer land in freed memory."""
        if self._closed:
            return
        self.drain()
        self._closed = True
        for s in self.slots:
            s.host = None
            if hasattr(self.backend, "free_event"):
                for ev in (s.uploaded, s.done):
                    try:
                        self.backend.free_event(ev)
                    except Exception:
                        pass
            s.uploaded = s.done = None
        if hasattr(self.backend, "close"):
            self.backend.close()
