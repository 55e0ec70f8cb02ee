#!/usr/bin/env python3
"""Frames that arrive in HOST memory (a CPU decoder, a capture card, a socket) -> GPU operator, with the upload hidden:
the ingest ring of `BatchedFramePipeline` keeps K slots of pinned host memory and device surfaces in flight, copies slot i+1
on a copy stream while slot i is processed on the task's stream, and hands finished slots back in order.

    python examples/host_fed_pipeline.py [raw_nv12_file width height] [--op convert|resize|ud|preproc]

Without a file it streams synthetic 1080p frames.  One process drives one GPU; for several GPUs start one process per GPU
(rank r takes frames shard_frames(total, r, world)) -- no frame crosses xGMI, the only collective is the 32-byte coefficient
broadcast (see bench.py)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import python_vali as vali  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    op = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--op=")), "preproc")
    gpu, per_slot = 0, 8
    vali.pipeline.bind_to_gpu_numa(gpu)                    # stay on the CPUs next to the GPU
    if len(args) == 3:
        path, w, h = args[0], int(args[1]), int(args[2])
        raw = np.memmap(path, np.uint8, "r")
        frame_bytes = w * h * 3 // 2
        nframes = raw.size // frame_bytes
        chunks = (raw[i * frame_bytes:(i + per_slot) * frame_bytes] for i in range(0, nframes, per_slot))
    else:
        w, h, nframes = 1920, 1080, 512
        frame_bytes = w * h * 3 // 2
        one = np.random.default_rng(0).integers(16, 236, frame_bytes * per_slot, dtype=np.uint8)
        chunks = (one for _ in range(nframes // per_slot))
    dst_size = None if op == "convert" else (640, 384) if op == "preproc" else (w // 2, h // 2)
    pipe = vali.BatchedFramePipeline(gpu, w, h, per_slot, vali.PixelFormat.RGB, op=op, dst_size=dst_size)
    pipe.set_coefficients(vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG))
    ring = pipe.ingest_ring(slots=3, frames_per_slot=per_slot)
    t0, done, checksum = time.perf_counter(), 0, 0
    import torch
    for tag, dsts in ring.feed(chunks):
        # the consumer: whatever comes next (a network, an encoder).  Here: touch the first output through DLPack, no copy.
        # Everything must be read BEFORE the generator is resumed -- the slot is reused then.
        t = torch.from_dlpack(dsts[0])
        checksum += int(t.view(torch.uint8).reshape(-1)[:4096].sum().item()) if t.dtype == torch.uint8 else int(t.reshape(-1)[:1024].abs().sum().item())
        done += per_slot
    dt = time.perf_counter() - t0
    ring.close()
    print(f"{op}: {done} frames {w}x{h} -> {pipe.dst_format.name} {pipe.dst_size[0]}x{pipe.dst_size[1]} in {dt * 1e3:.1f} ms = "
          f"{done / dt:.0f} frames/s, {done * frame_bytes / dt / 1e9:.1f} GB/s over PCIe (checksum {checksum})")


if __name__ == "__main__":
    main()
