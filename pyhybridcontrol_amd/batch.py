"""Batched harness: many (agent, scenario) MILP instances per call, sharded over ranks.

Instances are independent: flatten to i = s * n_agents + a, give rank g the contiguous
block [g*B/W, (g+1)*B/W); no collective is needed to compute, one all-gather returns the results.
The gather runs over RCCL (libmldgpu mld_gather, one process per GPU) or, for CPU tests of the
sharding logic, over any object with an ``all_gather(array) -> list`` method (torch.distributed/gloo).
"""
import numpy as np

from . import gpu, host, _lib


def shard_range(total, rank, world):
    """contiguous, balanced: the first (total % world) ranks get one extra instance"""
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def flatten_instances(n_agents, n_scen):
    """instance i = s * n_agents + a  ->  (model_idx, scenario_idx) arrays"""
    i = np.arange(n_agents * n_scen)
    return (i % n_agents).astype(np.int32), (i // n_agents).astype(np.int64)


class RcclGather(object):
    """all-gather of float64 arrays through libmldgpu's RCCL communicator"""

    def __init__(self, world, rank, unique_id_bytes):
        import ctypes as C
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(bytes(unique_id_bytes))
        _lib.check(_lib.load().mld_comm_init(int(world), int(rank), buf))
        self.world = world

    @staticmethod
    def unique_id():
        import ctypes as C
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        _lib.check(_lib.load().mld_comm_unique_id(buf))
        return bytes(buf)

    def all_gather(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        out = np.zeros((self.world,) + arr.shape)
        _lib.check(_lib.load().mld_gather(_lib.dptr(arr), int(arr.size), _lib.dptr(out)))
        return list(out)

    def gather_results(self, problem):
        """(objective, status, step-0 slice) of every rank's last solve, straight from the device buffers (mld_gather_results):
        array (world, batch, 2 + nv)"""
        import ctypes as C
        w = 2 + problem.model.nv
        out = np.zeros((self.world, problem.batch, w))
        wo = C.c_int()
        _lib.check(_lib.load().mld_gather_results(problem._h, _lib.dptr(out), C.byref(wo)))
        assert wo.value == w
        return out


class TcpRendezvous(object):
    """Launcher side channel without torch: rank / world / address from the environment (RANK, WORLD_SIZE, MASTER_ADDR,
    MASTER_PORT as `python -m torch.distributed.run` exports them -- nothing else listens on that port when torch.distributed
    is not initialised).  Rank 0 accepts one connection per peer; every operation is an all-gather of small byte blobs through
    rank 0, executed in lockstep.  Enough for the 128 bytes of the RCCL unique id and the barrier / max-over-ranks of a
    benchmark; the data path never uses it."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=600.0):
        import os
        import socket
        import struct
        import time
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(port if port is not None else os.environ.get("MASTER_PORT", "29500"))
        # under torch.distributed.run the agent's own store listens on MASTER_PORT: the side channel takes the first free port
        # of a fixed candidate list above it, and a handshake (magic + launcher run id) tells peers they found the right one
        cands = [base] if port is not None else [base + k for k in (1, 101, 211, 307, 401)]
        magic = b"MLDRDZV1" + os.environ.get("TORCHELASTIC_RUN_ID", "none").encode()[:32]
        self.conns = {}
        self.sock = None
        if self.world == 1:
            return
        t0 = time.time()
        if self.rank == 0:
            srv = None
            while srv is None:
                for cand in cands:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    try:
                        srv.bind(("", cand))
                        break
                    except OSError:
                        srv.close()
                        srv = None
                if srv is None:
                    if time.time() - t0 > timeout:
                        raise OSError("rendezvous: no free port among %s" % cands)
                    time.sleep(0.2)
            srv.listen(self.world)
            srv.settimeout(timeout)
            while len(self.conns) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(timeout)
                try:
                    hello = self._recvn(c, len(magic) + 4)
                except (ConnectionError, OSError):
                    c.close()
                    continue
                if hello[:len(magic)] != magic:
                    c.close()
                    continue
                c.sendall(b"OK" + magic)
                self.conns[struct.unpack("i", hello[len(magic):])[0]] = c
            srv.close()
        else:
            while self.sock is None:
                for cand in cands:
                    try:
                        c = socket.create_connection((addr, cand), timeout=2.0)
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.settimeout(5.0)
                        c.sendall(magic + struct.pack("i", self.rank))
                        if self._recvn(c, 2 + len(magic)) == b"OK" + magic:
                            c.settimeout(timeout)
                            self.sock = c
                            break
                        c.close()
                    except (OSError, ConnectionError):
                        pass
                if self.sock is None:
                    if time.time() - t0 > timeout:
                        raise TimeoutError("rendezvous: rank 0 not found on %s ports %s" % (addr, cands))
                    time.sleep(0.1)

    @staticmethod
    def _recvn(c, k):
        buf = b""
        while len(buf) < k:
            part = c.recv(k - len(buf))
            if not part:
                raise ConnectionError("rendezvous peer closed the connection")
            buf += part
        return buf

    @classmethod
    def _send_blob(cls, c, blob):
        import struct
        c.sendall(struct.pack("q", len(blob)) + blob)

    @classmethod
    def _recv_blob(cls, c):
        import struct
        return cls._recvn(c, struct.unpack("q", cls._recvn(c, 8))[0])

    def all_gather_bytes(self, blob):
        blob = bytes(blob)
        if self.world == 1:
            return [blob]
        if self.rank == 0:
            parts = [blob] + [self._recv_blob(self.conns[r]) for r in range(1, self.world)]
            for r in range(1, self.world):
                for part in parts:
                    self._send_blob(self.conns[r], part)
            return parts
        self._send_blob(self.sock, blob)
        return [self._recv_blob(self.sock) for _ in range(self.world)]

    def broadcast(self, blob, src=0):
        return self.all_gather_bytes(blob if self.rank == src else b"")[src]

    def barrier(self):
        self.all_gather_bytes(b"1")

    def all_max(self, value):
        import struct
        return max(struct.unpack("d", b)[0] for b in self.all_gather_bytes(struct.pack("d", float(value))))

    def close(self):
        if self.world > 1:
            self.barrier()
        for c in self.conns.values():
            c.close()
        if self.sock is not None:
            self.sock.close()
        self.conns, self.sock = {}, None


class TorchGather(object):
    """same interface over torch.distributed (gloo on CPU, nccl=RCCL on GPU) -- used by the CPU tests"""

    def __init__(self, dist):
        self.dist = dist
        self.world = dist.get_world_size()

    def all_gather(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
        if self.dist.get_backend() == "nccl":            # RCCL moves device buffers only
            t = t.cuda()
        outs = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [o.cpu().numpy() for o in outs]


def gather_sharded(local, total, rank, world, gatherer):
    """local: (n_local, k) rows of this rank's block -> (total, k) on every rank.  Blocks may differ by one
    row, so they are padded to the largest block for the fixed-size all-gather."""
    local = np.asarray(local, dtype=np.float64)
    if local.ndim == 1:
        local = local[:, None]
    cap = -(-total // world)
    pad = np.zeros((cap, local.shape[1]))
    pad[:local.shape[0]] = local
    parts = gatherer.all_gather(pad)
    out = np.zeros((total, local.shape[1]))
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        out[lo:hi] = parts[r][:hi - lo]
    return out


class BatchSolver(object):
    """n_agents models x n_scenarios instances on this rank's GPU"""

    def __init__(self, mats_list, dims, atoms_list, N_p, N_tilde, **opts):
        self.model = gpu.GpuModel(mats_list, dims)
        cost = host.stack_costs([host.cost_from_atoms(a, dims, N_p, N_tilde) for a in atoms_list])
        self.problem = gpu.GpuProblem(self.model, N_p, N_tilde, cost, **opts)
        self.n_agents = len(mats_list)

    def set_atoms(self, atoms_list):
        d = self.model.dims
        self.problem.set_cost(host.stack_costs([host.cost_from_atoms(a, d, self.problem.N_p, self.problem.N_tilde)
                                                for a in atoms_list]))

    def solve(self, x0, omega, model_idx=None):
        return self.problem.solve(x0, omega, model_idx)

    def close(self):
        self.problem.close()
        self.model.close()
