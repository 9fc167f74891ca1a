"""N>1 path on CPU: contiguous sharding + result gather with world_size 2 over gloo (127.0.0.1)."""
import os
import sys

import numpy as np
import pytest

from pyhybridcontrol_amd.batch import shard_range, flatten_instances, gather_sharded, TorchGather


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 8, 4096 * 64, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    mi, si = flatten_instances(64, 3)
    assert mi[:65].tolist() == list(range(64)) + [0] and si[63] == 0 and si[64] == 1


def _worker(rank, world, port, total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    # a rank's "results": row i holds (i, i^2, status) -- depends only on the global instance id
    idx = np.arange(lo, hi, dtype=np.float64)
    local = np.stack([idx, idx ** 2, idx % 3], axis=1)
    full = gather_sharded(local, total, rank, world, TorchGather(dist))
    dist.barrier()
    q.put((rank, full))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [9, 64])
def test_gather_sharded_world2_gloo(total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + total) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    idx = np.arange(total, dtype=np.float64)
    expect = np.stack([idx, idx ** 2, idx % 3], axis=1)
    for r in range(2):
        assert np.array_equal(res[r], expect)


class _RdzvGather(object):
    """all_gather(array) over the torch-free rendezvous -- stands in for the RCCL gather on CPU"""

    def __init__(self, rdzv):
        self.rdzv, self.world = rdzv, rdzv.world

    def all_gather(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        return [np.frombuffer(b, dtype=np.float64).reshape(arr.shape) for b in self.rdzv.all_gather_bytes(arr.tobytes())]


def _rdzv_worker(rank, world, port, total, q):
    from pyhybridcontrol_amd.batch import TcpRendezvous
    rd = TcpRendezvous(rank=rank, world=world, addr="127.0.0.1", port=port, timeout=60.0)
    uid = rd.broadcast(bytes(range(128)) if rank == 0 else b"", src=0)      # the 128 id bytes of mld_comm_unique_id
    rd.barrier()
    slowest = rd.all_max(10.0 + rank)
    lo, hi = shard_range(total, rank, world)
    idx = np.arange(lo, hi, dtype=np.float64)
    full = gather_sharded(np.stack([idx, idx ** 2], axis=1), total, rank, world, _RdzvGather(rd))
    rd.close()
    q.put((rank, uid, slowest, full))


def test_tcp_rendezvous_world3_without_torch():
    """the launcher side channel of bench.py --gpus N: unique-id broadcast, barrier, max over ranks, uneven shards"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    world, total = 3, 10
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    idx = np.arange(total, dtype=np.float64)
    for rank, uid, slowest, full in res:
        assert uid == bytes(range(128))
        assert slowest == 10.0 + world - 1
        assert np.array_equal(full, np.stack([idx, idx ** 2], axis=1))
