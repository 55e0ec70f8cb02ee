"""N>1 path of the frame-sharded pipeline on CPU: world_size 2, gloo backend.

Covers what bench.py does across ranks: shard_frames() partitions the global batch with no
overlap, and the one collective -- broadcast of the 32-byte coefficient block from rank 0 --
delivers bit-identical coefficients to every rank."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vali_amd as vali
    from vali_amd import tasks

    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    coeffs = tasks._nv12_variant(cc) if rank == 0 else None      # only rank 0 knows them
    got = vali.broadcast_coefficients(coeffs, src=0)
    begin, end = vali.shard_frames(4096, rank, world)
    dist.barrier()
    q.put((rank, got, (begin, end)))
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    import torch.multiprocessing as mp

    from vali_amd import tasks

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = tuple(float(np.float32(c)) for c in tasks.CSC_NPP_709CSC)
    assert results[0][1] == want and results[1][1] == want
    assert results[0][2] == (0, 2048) and results[1][2] == (2048, 4096)


@pytest.mark.parametrize("total,world", [(4096, 8), (10, 4), (7, 8), (0, 2), (512, 1)])
def test_shard_frames_partitions_exactly(total, world):
    import vali_amd as vali

    spans = [vali.shard_frames(total, r, world) for r in range(world)]
    covered = [i for b, e in spans for i in range(b, e)]
    assert covered == list(range(total))
    assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= -(-total // world)


# ---- the ingest ring (host-fed pipeline) without a GPU: its ordering logic against a model of two asynchronous streams ----
class _FakeStreams:
    """Backend of IngestRing with the device replaced by a model: two FIFO streams of deferred operations that only run
    when the host waits for / polls an event -- like a GPU, nothing has happened just because it was enqueued.  Uploads
    read the pinned buffer when they EXECUTE, so a buffer refilled too early corrupts the frames (the bug class the ring's
    acquire / reap protocol exists to prevent)."""

    def __init__(self):
        self.copy_stream, self.task_stream = "copy", "task"
        self.q = {"copy": [], "task": []}
        self.fired = set()
        self.n_events = 0
        self.max_pending = 0

    def host_buffer(self, nbytes):
        return np.zeros(nbytes, np.uint8)

    def new_event(self):
        self.n_events += 1
        return self.n_events

    def upload(self, host, frame_bytes, srcs):
        self.q["copy"].append(("run", lambda: [s.__setitem__(slice(None), host[i * frame_bytes:(i + 1) * frame_bytes])
                                               for i, s in enumerate(srcs)]))

    def record(self, event, stream):
        self.fired.discard(event)             # hipEventRecord: the event now stands for THIS point of the stream
        self.q[stream].append(("fire", event))

    def stream_wait(self, stream, event):
        self.q[stream].append(("wait", event))

    def _advance(self, budget):
        """run up to `budget` operations, respecting waits; returns how many ran"""
        ran = 0
        while ran < budget:
            progressed = False
            for name in ("copy", "task"):
                if not self.q[name]:
                    continue
                kind, arg = self.q[name][0]
                if kind == "wait" and arg not in self.fired:
                    continue
                self.q[name].pop(0)
                if kind == "run":
                    arg()
                elif kind == "fire":
                    self.fired.add(arg)
                ran += 1
                progressed = True
                break
            if not progressed:
                break
        return ran

    def host_wait(self, event):
        while event not in self.fired:
            assert self._advance(1) == 1, "deadlock: the event can never fire"

    def is_done(self, event):
        self._advance(1)                      # the device makes a little progress between two polls
        return event in self.fired


class _FakePipe:
    """what IngestRing needs of a BatchedFramePipeline: geometry, a stream, batches, a launch -- on numpy arrays"""

    def __init__(self, backend, width, height, key):
        self.gpu_id, self.width, self.height, self.key, self.b = -1, width, height, key, backend

    Stream = "task"

    def _make_batch(self, n):
        fb = self.width * self.height * 3 // 2
        srcs = [np.zeros(fb, np.uint8) for _ in range(n)]
        dsts = [np.zeros(fb, np.uint8) for _ in range(n)]
        return srcs, dsts, (srcs, dsts)

    def _launch(self, batch):
        import vali_amd as vali
        srcs, dsts = batch
        self.b.q["task"].append(("run", lambda: [d.__setitem__(slice(None), s[::-1] ^ self.key) for s, d in zip(srcs, dsts)]))
        return True, vali.TaskExecInfo.SUCCESS


def _ring_run(frames, per_slot, slots, key, w=8, h=4):
    from vali_amd.pipeline import IngestRing
    be = _FakeStreams()
    ring = IngestRing(_FakePipe(be, w, h, key), slots=slots, frames_per_slot=per_slot, backend=be)
    chunks = [frames[i:i + per_slot] for i in range(0, len(frames), per_slot)]
    out = {}
    for tag, dsts in ring.feed([np.concatenate(c) for c in chunks]):
        out[tag] = [d.copy() for d in dsts[:len(chunks[tag])]]          # consumed before the slot can be reused
    assert sorted(out) == list(range(len(chunks)))
    return [f for tag in sorted(out) for f in out[tag]], ring


@pytest.mark.parametrize("slots,per_slot,nframes", [(2, 4, 20), (3, 4, 22), (4, 1, 9), (3, 16, 16)])
def test_ingest_ring_orders_uploads_and_launches(slots, per_slot, nframes):
    rng = np.random.default_rng(slots * 100 + nframes)
    frames = [rng.integers(0, 256, 8 * 4 * 3 // 2, dtype=np.uint8) for _ in range(nframes)]
    got, ring = _ring_run(frames, per_slot, slots, key=0x5a)
    assert len(got) == nframes
    for f, g in zip(frames, got):
        assert np.array_equal(g, f[::-1] ^ 0x5a)
    assert ring.frames_submitted == -(-nframes // per_slot) * per_slot
    with pytest.raises(ValueError):
        from vali_amd.pipeline import IngestRing
        IngestRing(_FakePipe(_FakeStreams(), 8, 4, 0), slots=1, backend=_FakeStreams())


def test_ingest_ring_refuses_a_slot_whose_outputs_nobody_took():
    from vali_amd.pipeline import IngestRing
    be = _FakeStreams()
    ring = IngestRing(_FakePipe(be, 8, 4, 1), slots=2, frames_per_slot=1, backend=be)
    for _ in range(2):
        ring.submit(ring.acquire())
    with pytest.raises(RuntimeError, match="reap"):
        ring.acquire()
    assert len(ring.reap()) >= 1 and ring.acquire() is ring.slots[0]


def test_ingest_ring_short_last_chunk_and_close():
    """a final chunk shorter than a slot yields only its VALID outputs (the slot's tail still holds an earlier
    chunk's frames); close() drops the views into the pinned buffers and the events, and a closed ring refuses work."""
    from vali_amd.pipeline import IngestRing
    be = _FakeStreams()
    freed = []
    be.free_event = freed.append
    ring = IngestRing(_FakePipe(be, 8, 4, 0x11), slots=2, frames_per_slot=4, backend=be)
    rng = np.random.default_rng(3)
    frames = [rng.integers(0, 256, 48, dtype=np.uint8) for _ in range(10)]          # 4 + 4 + 2
    chunks = [np.concatenate(frames[i:i + 4]) for i in range(0, 10, 4)]
    sizes = {tag: len(dsts) for tag, dsts in ring.feed(chunks)}
    assert sizes == {0: 4, 1: 4, 2: 2}
    with pytest.raises(ValueError):
        list(ring.feed([np.zeros(47, np.uint8)]))                                     # not whole frames
    # a fill callback reports what it wrote
    got = list(ring.feed([frames[0]], fill=lambda host, item: (host.__setitem__(slice(0, item.size), item), 1)[1]))
    assert [len(d) for _t, d in got] == [1]
    # a bad chunk behind good ones: the good ones are still delivered before the error; fill() == 0 ends the input
    seen = []
    with pytest.raises(ValueError):
        for tag, dsts in ring.feed([chunks[0], chunks[1], np.zeros(47, np.uint8)]):
            seen.append((tag, len(dsts)))
    assert seen == [(0, 4), (1, 4)]
    state = {"n": 0}
    def two_then_end(host, item):
        state["n"] += 1
        if state["n"] > 2:
            return 0
        host[:item.size] = item
        return 4
    assert [(t, len(d)) for t, d in ring.feed([chunks[0]] * 5, fill=two_then_end)] == [(0, 4), (1, 4)]
    with pytest.raises(TypeError):
        list(ring.feed([chunks[0]], fill=lambda host, item: True))
    assert [(t, len(d)) for t, d in ring.feed(chunks)] == [(0, 4), (1, 4), (2, 2)]       # the ring is still usable, no slot was lost
    ring.close()
    assert all(s.host is None and s.uploaded is None and s.done is None for s in ring.slots) and len(freed) == 4
    with pytest.raises(RuntimeError, match="closed"):
        ring.acquire()
    ring.close()                                                                      # idempotent


def _ring_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vali_amd as vali
    from vali_amd import tasks

    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    coeffs = vali.broadcast_coefficients(tasks._nv12_variant(cc) if rank == 0 else None, src=0)
    key = int(abs(coeffs[0]) * 100) & 0xff                     # every rank derives the same operator from the broadcast
    total = 37                                                # global batch, not a multiple of anything
    begin, end = vali.shard_frames(total, rank, world)
    frames = [np.random.default_rng(1000 + i).integers(0, 256, 48, dtype=np.uint8) for i in range(begin, end)]
    got, _ = _ring_run(frames, per_slot=4, slots=3, key=key)
    bad = sum(not np.array_equal(g, f[::-1] ^ key) for f, g in zip(frames, got))
    t = torch.tensor([sum(int(g.astype(np.uint64).sum()) for g in got), bad, len(got)], dtype=torch.int64)
    dist.all_reduce(t)                                         # checksum of checksums, mismatches, frames: the parity report
    q.put((rank, key, t.tolist()))
    dist.destroy_process_group()


def test_ingest_ring_world2_gloo():
    import torch.multiprocessing as mp
    from vali_amd import tasks

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    key = int(abs(float(np.float32(tasks.CSC_NPP_709CSC[0]))) * 100) & 0xff
    want = sum(int((np.random.default_rng(1000 + i).integers(0, 256, 48, dtype=np.uint8)[::-1] ^ key).astype(np.uint64).sum())
               for i in range(37))
    for rank, k, (checksum, bad, n) in results:
        assert (k, checksum, bad, n) == (key, want, 0, 37)


# ---- host placement and operator geometry (no GPU) ---------------------------------------------------------------
def test_gpu_numa_cpus_reads_sysfs(tmp_path):
    from vali_amd.pipeline import gpu_numa_cpus, bind_to_gpu_numa, _parse_cpulist
    dev = tmp_path / "bus/pci/devices/0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices/system/node/node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("64-127,192-255\n")
    assert gpu_numa_cpus("0000:C1:00.0", str(tmp_path)) == list(range(64, 128)) + list(range(192, 256))
    assert gpu_numa_cpus("c1:00.0", str(tmp_path))[0] == 64                 # domain-less form
    (dev / "numa_node").write_text("-1\n")
    assert gpu_numa_cpus("0000:c1:00.0", str(tmp_path)) is None             # single-node host
    assert gpu_numa_cpus("0000:99:00.0", str(tmp_path)) is None
    assert _parse_cpulist("0-2,5") == [0, 1, 2, 5]
    # binding: only CPUs this process may use, and never an error when the topology is unknown
    (dev / "numa_node").write_text("1\n")
    mine = sorted(os.sched_getaffinity(0))
    (node / "cpulist").write_text(f"{mine[0]}\n")
    try:
        assert bind_to_gpu_numa(0, str(tmp_path), pci_bus_id="0000:c1:00.0") == [mine[0]]
        assert sorted(os.sched_getaffinity(0)) == [mine[0]]
    finally:
        os.sched_setaffinity(0, mine)
    assert bind_to_gpu_numa(0, str(tmp_path), pci_bus_id="0000:99:00.0") is None


def test_gpu_clock_state_reads_sysfs(tmp_path):
    """The telemetry next to the bench line (bench.py host_placement.gpu_state): amdgpu's sysfs files, every field None
    when the platform does not say, never an error."""
    from vali_amd.pipeline import gpu_clock_state, sample_clocks_under_load
    dev = tmp_path / "bus/pci/devices/0000:c1:00.0"
    (dev / "hwmon/hwmon3").mkdir(parents=True)
    (dev / "pp_dpm_sclk").write_text("0: 132Mhz\n1: 1730Mhz *\n2: 2400Mhz\n")
    (dev / "pp_dpm_mclk").write_text("0: 900Mhz\n1: 2000Mhz *\n")
    (dev / "current_compute_partition").write_text("SPX\n")
    (dev / "hwmon/hwmon3/power1_cap").write_text("1400000000\n")
    (dev / "hwmon/hwmon3/power1_average").write_text("912000000\n")
    (dev / "hwmon/hwmon3/temp1_input").write_text("61000\n")
    (dev / "hwmon/hwmon3/temp2_input").write_text("74000\n")
    st = gpu_clock_state("0000:C1:00.0", str(tmp_path))
    assert (st["sclk_mhz"], st["mclk_mhz"], st["fclk_mhz"]) == (1730, 2000, None)
    assert (st["power_cap_w"], st["power_w"], st["temp_c"]) == (1400.0, 912.0, 74.0)
    assert st["compute_partition"] == "SPX" and st["memory_partition"] is None
    # no pp_dpm files: the hwmon frequency inputs (Hz) serve
    (dev / "pp_dpm_sclk").unlink()
    (dev / "hwmon/hwmon3/freq1_input").write_text("2100000000\n")
    assert gpu_clock_state("c1:00.0", str(tmp_path))["sclk_mhz"] == 2100
    # unknown device: all None
    assert all(v is None for v in gpu_clock_state("0000:99:00.0", str(tmp_path)).values())
    # under load: the step runs at least once, clocks are reported as [min, max]
    calls = []

    def step():
        calls.append(1)
        (dev / "pp_dpm_mclk").write_text(f"0: 900Mhz\n1: {1900 + 100 * (len(calls) % 2)}Mhz *\n")
    rep = sample_clocks_under_load("0000:c1:00.0", step, seconds=0.02, sysfs=str(tmp_path))
    assert calls and rep["under_load"]["samples"] == len(calls)
    assert rep["under_load"]["mclk_mhz"][0] in (1900, 2000) and rep["under_load"]["mclk_mhz"][1] in (1900, 2000)
    assert rep["power_cap_w"] == 1400.0 and rep["before"]["mclk_mhz"] == 2000


def test_op_geometry():
    import vali_amd as vali
    from vali_amd.pipeline import op_geometry
    assert op_geometry("convert", 3840, 2160, vali.RGB) == (vali.RGB, (3840, 2160), 37324800)
    assert op_geometry("resize", 3840, 2160, vali.RGB, (1920, 1088)) == (vali.NV12, (1920, 1088), 12441600 + 3133440)
    assert op_geometry("ud", 3840, 2160, vali.RGB, (1920, 1080)) == (vali.RGB, (1920, 1080), 18662400)
    assert op_geometry("preproc", 1920, 1080, vali.RGB, (640, 384))[2] == 3110400 + 640 * 384 * 12
    with pytest.raises(ValueError):
        op_geometry("rotate", 64, 48, vali.RGB)
