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
