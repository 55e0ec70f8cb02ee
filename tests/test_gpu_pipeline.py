"""The sharded pipeline on a real GPU: every operator (`op`) gives the bytes of the task it wraps, and the host-fed ring
(pinned staging, H2D on a copy stream overlapped with the operator) delivers every frame once, in order, bit-exact -- with
the producer refilling a slot's pinned buffer while older slots are still in flight."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def nv12_frames(n, w, h, seed):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8) for _ in range(n)]


def download(vali, gpu, surf, stream=None):
    out = np.zeros(surf.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(gpu, stream).Run(surf, out)[0]
    return out


@pytest.mark.parametrize("op,dst_size", [("convert", None), ("resize", (320, 184)), ("ud", (320, 180)), ("preproc", (224, 160))])
def test_pipeline_ops_equal_their_tasks(vali, gpu, oracle, op, dst_size):
    w, h, n = 640, 360, 6
    frames = nv12_frames(n, w, h, 3)
    pipe = vali.BatchedFramePipeline(gpu, w, h, n, vali.RGB, op=op, dst_size=dst_size)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    pipe.set_coefficients(cc)
    up = vali.PyFrameUploader(gpu, pipe.Stream)
    for f, s in zip(frames, pipe.srcs):
        assert up.Run(f, s)[0]
    assert pipe.run() == (True, vali.TaskExecInfo.SUCCESS)
    got = [download(vali, gpu, d, pipe.Stream) for d in pipe.dsts]
    # the same through the plain single-surface task
    if op == "convert":
        task, fmt, size = vali.PySurfaceConverter(gpu), vali.RGB, (w, h)
    elif op == "resize":
        task, fmt, size = vali.PySurfaceResizer(vali.NV12, gpu), vali.NV12, dst_size
    elif op == "ud":
        task, fmt, size = vali.PySurfaceUD(gpu), vali.RGB, dst_size
    else:
        task = vali.PySurfacePreprocessor(gpu, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), div=255.0)
        fmt, size = vali.RGB_32F_PLANAR, dst_size
    assert (pipe.dst_format, pipe.dst_size) == (fmt, tuple(size))
    for f, g in zip(frames, got):
        src, dst = vali.Surface.Make(vali.NV12, w, h, gpu), vali.Surface.Make(fmt, size[0], size[1], gpu)
        assert vali.PyFrameUploader(gpu).Run(f, src)[0]
        ok = task.Run(src, dst, cc) if op in ("convert", "preproc") else task.Run(src, dst)
        assert ok[0]
        assert np.array_equal(g, download(vali, gpu, dst))
    if op == "convert":
        from vali_amd import tasks
        want = oracle.nv12_to_rgb(frames[0].reshape(h * 3 // 2, w), w, h, oracle.csc_from_tuple(tasks.CSC_NPP_709CSC), "RGB")
        assert np.array_equal(got[0], want.reshape(-1))


@pytest.mark.parametrize("slots,per_slot,nframes", [(2, 3, 17), (3, 4, 40), (4, 1, 9)])
def test_ingest_ring_delivers_every_frame_once_in_order(vali, gpu, oracle, slots, per_slot, nframes):
    from vali_amd import tasks
    w, h = 256, 96
    frames = nv12_frames(nframes, w, h, slots * 10 + per_slot)
    pipe = vali.BatchedFramePipeline(gpu, w, h, per_slot, vali.RGB)
    pipe.set_coefficients(vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG))
    ring = pipe.ingest_ring(slots=slots, frames_per_slot=per_slot)
    chunks = [frames[i:i + per_slot] for i in range(0, nframes, per_slot)]
    out = {}
    for tag, dsts in ring.feed([np.concatenate(c) for c in chunks]):
        out[tag] = [download(vali, gpu, d, pipe.Stream) for d in dsts[:len(chunks[tag])]]   # consumed before the slot is reused
    ring.close()
    assert sorted(out) == list(range(len(chunks))) and ring.frames_submitted == len(chunks) * per_slot
    k = oracle.csc_from_tuple(tasks.CSC_NPP_709CSC)
    flat = [g for tag in sorted(out) for g in out[tag]]
    assert len(flat) == nframes
    for f, g in zip(frames, flat):
        assert np.array_equal(g, oracle.nv12_to_rgb(f.reshape(h * 3 // 2, w), w, h, k, "RGB").reshape(-1))
    with pytest.raises(RuntimeError, match="reap"):        # a slot whose outputs nobody took is not handed out again
        r2 = pipe.ingest_ring(slots=2, frames_per_slot=per_slot)
        try:
            for _ in range(2):
                r2.submit(r2.acquire())
            r2.acquire()
        finally:
            r2.close()
