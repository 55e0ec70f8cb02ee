"""Batched, frame-sharded conversion pipeline: one process per GPU.

New design (the reference has no multi-GPU code: SURVEY.md 2.1 / 8e).  Frames are
independent, so N frames are split into contiguous blocks, one per rank; no frame ever
crosses xGMI.  The only exchange step is a broadcast of the 32-byte colour-coefficient
block (vali_csc) from rank 0 over torch.distributed (backend "nccl" = RCCL on ROCm;
"gloo" in the CPU tests), so that every GPU converts with the same matrix.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np

from . import tasks
from .enums import ColorspaceConversionContext, PixelFormat, TaskExecInfo
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


class BatchedFramePipeline:
    """Owns this rank's shard of surfaces and converts it with one launch per step."""

    def __init__(self, gpu_id: int, width: int, height: int, frames: int,
                 dst_format: PixelFormat = PixelFormat.RGB, stream=None):
        from .tasks import PySurfaceConverter

        self.gpu_id, self.width, self.height, self.frames = gpu_id, width, height, frames
        self.converter = PySurfaceConverter(gpu_id, stream)
        self.srcs = [Surface.Make(PixelFormat.NV12, width, height, gpu_id) for _ in range(frames)]
        self.dsts = [Surface.Make(dst_format, width, height, gpu_id) for _ in range(frames)]
        self.batch = self.converter.PrepareBatch(self.srcs, self.dsts)
        self._csc = None

    @property
    def Stream(self) -> int:
        return self.converter.Stream

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
        return coeffs

    def run_async(self) -> Tuple[bool, TaskExecInfo]:
        if self._csc is None:
            raise RuntimeError("set_coefficients() first")
        return self.converter.RunBatchAsync(self.batch, csc=self._csc)

    def run(self) -> Tuple[bool, TaskExecInfo]:
        r = self.run_async()
        self.converter._sync()
        return r
