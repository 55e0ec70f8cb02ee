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
