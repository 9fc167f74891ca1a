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


class _StubProblem(object):
    """stands in for a GpuProblem handle: launch / finish take a rank- and handle-dependent time, calls are logged"""

    def __init__(self, rank, hid, log):
        self.rank, self.hid, self.log, self.step, self.set = rank, hid, log, -1, -1

    def select(self, k):
        self.set = k
        self.log.append(("select", self.hid, k))

    def launch(self):
        self.step += 1
        self.log.append(("launch", self.hid, self.set))

    def finish(self):
        import time
        time.sleep(0.002 * ((self.rank * 7 + self.hid * 3 + self.step) % 5))     # ranks and handles finish at different speeds
        self.log.append(("finish", self.hid, self.set))
        return dict(n_optimal=1, set=self.set)


class _StubGather(object):
    """the result gather as a tagged collective over the rendezvous: every rank must arrive with the same (handle, scenario set)"""

    def __init__(self, rd, log):
        self.rd, self.log = rd, log

    def gather_results(self, prob):
        tag = ("%d:%d" % (prob.hid, prob.set)).encode()
        got = self.rd.all_gather_bytes(tag)
        self.log.append(("gather", prob.hid, prob.set, tuple(got)))
        assert all(g == tag for g in got), (tag, got)


def _pipeline_worker(rank, world, port, q):
    import bench
    from pyhybridcontrol_amd.batch import TcpRendezvous
    rd = TcpRendezvous(rank=rank, world=world, addr="127.0.0.1", port=port, timeout=60.0)
    log = []
    probs = [_StubProblem(rank, h, log) for h in range(2)]
    state = dict(k=0)
    res = bench.run_pipelined(probs, 7, state, 5, _StubGather(rd, log))
    rd.barrier()
    rd.close()
    q.put((rank, [r["set"] for r in res], [e[:3] for e in log]))


def test_two_ranks_two_handles_issue_the_same_collective_sequence():
    """bench.py's pipelined timed region (launch / finish / gather alternating over two handles) must issue its
    collectives in the same order on every rank whatever the ranks' solve times, or the first 8-GPU run deadlocks in RCCL"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r, (sets, log)) for r, sets, log in (q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] == [k % 5 for k in range(7)], "step j solves scenario set j, in order"
    assert res[0][1] == res[1][1], "both ranks walk the same sequence of selects, launches, finishes and gathers"
    kinds = [e[0] for e in res[0][1]]
    assert kinds.count("gather") == 7 and kinds.count("launch") == 7 and kinds.count("finish") == 7
    first_gather = kinds.index("gather")
    assert kinds[:first_gather].count("launch") == 2, "two solves are in flight before the first finish: the steps overlap"
