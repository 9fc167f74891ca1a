#!/usr/bin/env python3
"""Benchmark of the MPC hot path on MI355X:  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "MPC steps/sec ... 64-agent microgrid N=24", configs[3]): 64 distinct agents
(cluster of 7 water heaters + grid tie, N_p=24 -> n=575 variables / 200 binaries / 500 rows each) x scenarios;
the full configuration is 4096 scenarios x 64 agents over 8 GPUs, i.e. 512 scenarios x 64 agents = 32768
independent MILP instances per GPU.  WEAK scaling: every rank solves that per-GPU shard, instance ids are
contiguous blocks of the flattened (scenario, agent) index.  One "step" = one pass of the hot path
over the rank's resident batch: K3 (right-hand sides) + K5/K6 (cut-and-branch) with inputs already in HBM,
followed by the RCCL gather of (objective, status) when N > 1.  One "MPC step" = one agent-solve.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel k_solve) and
`cpu_baseline` (the C oracle on a bounded sample, host cores of this box) added.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--agents", type=int, default=64)
    ap.add_argument("--scenarios", type=int, default=512, help="scenarios per GPU (4096 over 8 GPUs)")
    ap.add_argument("--mip-gap", type=float, default=1e-2, help="relative MIP gap; 1e-2 is the reference's own setting "
                    "(micro_grid_control_simulation.py:232 MIPGap=1e-2)")
    ap.add_argument("--node-limit", type=int, default=400, help="per-instance node limit (stands in for the reference's TimeLimit)")
    ap.add_argument("--pivot-limit", type=int, default=20000, help="per-instance simplex iteration limit")
    ap.add_argument("--cpu-sample", type=int, default=24, help="instances timed with the CPU oracle (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--rehearse", action="store_true", help="multi-rank rehearsal on ONE GPU: gloo instead of RCCL, every rank on device 0")
    return ap.parse_args()


def make_shard(n_agents, n_scen, scen_offset):
    """agents (models + costs) are the same on every rank; scenarios are seeded by their GLOBAL index"""
    from pyhybridcontrol_amd import synthetic as syn
    cfg = syn.CONFIGS["cfg4"]
    n_h, N_p = cfg["n_h"], cfg["N_p"]
    N_t = N_p + 1
    agents = []
    for a in range(n_agents):
        rng = np.random.Generator(np.random.PCG64(cfg["seed"] * 1000 + a))
        mats, dims, params = syn.make_agent(n_h, rng)
        atoms = syn.make_cost(n_h, N_t, params)
        agents.append(dict(mats=mats, dims=dims, params=params, atoms=atoms))
    nx, nW = n_h, N_t * (n_h + 1)
    x0 = np.zeros((n_scen, n_agents, nx))
    om = np.zeros((n_scen, n_agents, nW))
    for s in range(n_scen):
        for a in range(n_agents):
            rng = np.random.Generator(np.random.PCG64([cfg["seed"], scen_offset + s, a]))
            xs, ws = syn.make_scenarios(n_h, N_t, 1, rng)
            x0[s, a], om[s, a] = xs[0], ws[0]
    # flatten i = s * n_agents + a  
    midx = np.tile(np.arange(n_agents, dtype=np.int32), n_scen)
    return agents, N_p, N_t, x0.reshape(-1, nx), om.reshape(-1, nW), midx


def cpu_baseline(agents, N_p, N_t, x0, om, midx, n_sample, gap, node_limit, pivot_limit):
    """the C oracle (oracle/mld_oracle.c, kind "port") on the first n_sample instances, one host thread"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import condense_np as cn
    import orc
    import tighten_np
    forms = {}
    t_total, n_done, n_opt = 0.0, 0, 0
    for i in range(min(n_sample, x0.shape[0])):
        a = int(midx[i])
        ag = agents[a]
        t0 = time.perf_counter()
        if a not in forms:   # condensing + tightening is per model; counted once like on the GPU (problem set-up)
            d = ag["dims"]
            tm = tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"])
            forms[a] = cn.standard_form(tm, ag["atoms"], N_p, N_t, nu_l=d["nu_l"])
            t0 = time.perf_counter()
        sf = forms[a]
        h = cn.rhs(sf["evo"], x0[i], om[i])
        q = cn.lin_cost(sf["cost"], x0[i], om[i])
        r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], gap_rel=gap, max_nodes=node_limit, presolve=0, max_pivots=pivot_limit)
        t_total += time.perf_counter() - t0
        n_done += 1
        n_opt += r["status"] == "optimal"
        if t_total > 40.0:
            break
    out = dict(value=round(n_done / t_total, 3), unit="agent-solves/s", cores=1, kind="port",
               sample="first %d instances of the rank-0 shard, same MIPGap/NodeLimit, oracle/mld_oracle.c single thread "
                      "(%d proven optimal)" % (n_done, n_opt))
    # an independent third-party CPU solver on the same instances (original, un-tightened rows), if scipy is there
    try:
        from scipy.optimize import milp, LinearConstraint, Bounds
        t_h, n_h = 0.0, 0
        raw = {}
        for i in range(min(8, n_done)):
            a = int(midx[i])
            ag = agents[a]
            if a not in raw:
                raw[a] = cn.standard_form(ag["mats"], ag["atoms"], N_p, N_t, nu_l=ag["dims"]["nu_l"])
            sf = raw[a]
            h = cn.rhs(sf["evo"], x0[i], om[i])
            q = cn.lin_cost(sf["cost"], x0[i], om[i])
            t0 = time.perf_counter()
            milp(q, constraints=LinearConstraint(sf["G"], -np.inf, h), bounds=Bounds(sf["lb"], sf["ub"]),
                 integrality=sf["is_bin"].astype(int), options=dict(mip_rel_gap=gap))
            t_h += time.perf_counter() - t0
            n_h += 1
        out["third_party"] = dict(solver="scipy.optimize.milp (HiGHS), mip_rel_gap=%g, 1 thread" % gap,
                                  value=round(n_h / t_h, 3), unit="agent-solves/s", sample="first %d instances" % n_h)
    except Exception as e:      # noqa: BLE001 -- reported, not fatal: the baseline above stands on its own
        out["third_party"] = dict(solver="scipy.optimize.milp (HiGHS)", error=str(e)[:200])
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.rehearse:
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world)
    from pyhybridcontrol_amd import _lib, gpu, host
    from pyhybridcontrol_amd.batch import RcclGather, TorchGather
    _lib.check(_lib.load().mld_set_device(local_rank))
    agents, N_p, N_t, x0, om, midx = make_shard(args.agents, args.scenarios, rank * args.scenarios)
    d = agents[0]["dims"]
    model = gpu.GpuModel([a["mats"] for a in agents], d)
    cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
    prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=args.mip_gap, max_nodes=args.node_limit, max_pivots=args.pivot_limit)
    n_local = x0.shape[0]
    prob.upload(x0, om, midx)                       # inputs resident in HBM before the timed region
    gatherer = None
    if world > 1:
        if args.rehearse:
            gatherer = TorchGather(dist)
        else:
            ids = [RcclGather.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            try:
                gatherer = RcclGather(world, rank, ids[0])
            except Exception as e:      # noqa: BLE001 -- every rank must take the same path: agreed below
                print("rank %d: libmldgpu communicator failed (%s)" % (rank, e), file=sys.stderr)
            import torch
            ok = torch.tensor([1 if gatherer is not None else 0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:     # the same RCCL all-gather through torch.distributed's communicator
                gatherer = TorchGather(dist)

    def sync():
        if world > 1:
            import torch
            dist.barrier()
            if not args.rehearse:
                torch.cuda.synchronize()

    def step():
        st = prob.solve_resident()                  # K3 + K5/K6 on resident inputs, HIP-event timed inside
        if gatherer is not None:                    # the trivial result gather (RCCL over xGMI)
            out = prob.download()
            gatherer.all_gather(np.stack([out["obj"], out["status"].astype(np.float64)], axis=1))
        return st

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- roofline of the dominant kernel (k_solve): bytes the rank-1 dictionary updates streamed / HIP-event time
    tel = prob.telemetry()
    out = prob.download()
    kernel_ms = float(np.mean([s["solve_ms"] for s in stats]))
    rows = int(tel["rows_updated"].sum())
    pivots = int(out["pivots"].sum())
    row_bytes = tel["row_bytes"]
    # bytes the simplex pivots stream: 64-byte sectors of the updated rows (read + write), the pivot row (read + write),
    # the cost row (read + write) and the multiplier column (one 64-byte sector per row: it is a strided gather)
    upd_bytes = 2.0 * rows * row_bytes + pivots * (2 * 8.0 * (prob.n + 1) + 2 * 8.0 * prob.n + 64.0 * prob.m)
    io_bytes = 8.0 * n_local * (prob.n + prob.m + d["nx"] + prob.nW)           # SURVEY 8d input/output minimum
    achieved = upd_bytes / (kernel_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_solve.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    lat = np.sort(tel["latency_ns"]) * 1e-6
    status = out["status"]
    fin = np.isfinite(out["obj"])
    with np.errstate(invalid="ignore"):
        gap = np.where(fin, (out["obj"] - out["lower_bound"]) / np.maximum(1e-9, np.abs(out["obj"])), np.nan)
    result = {
        "metric": "MPC steps/sec (whole node) + p50 solve latency, 64-agent microgrid N=24",
        "value": round(world * n_local * args.steps / elapsed, 2),
        "unit": "agent-solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE cfg4 shard: %d agents x %d scenarios per GPU (n_h=7, N_p=24: n=575, 200 binaries, m=500), "
                               "full branch-and-cut MILP, MIPGap=%g, NodeLimit=%d, IterationLimit=%d" % (args.agents, args.scenarios, args.mip_gap, args.node_limit, args.pivot_limit),
                   "instances_per_gpu": n_local, "microgrid_steps_per_s": round(world * n_local * args.steps / elapsed / args.agents, 3),
                   "p50_solve_latency_ms": round(float(lat[len(lat) // 2]), 3), "p99_solve_latency_ms": round(float(lat[int(len(lat) * 0.99)]), 3),
                   "status": {"optimal": int((status == 0).sum()), "infeasible": int((status == 1).sum()),
                              "node_limit": int((status == 2).sum()), "numerical": int((status == 3).sum())},
                   "no_incumbent": int((~fin).sum()), "median_gap_of_limited": (round(float(np.nanmedian(gap[status == 2])), 5) if (status == 2).any() else 0.0),
                   "nodes_per_instance": round(float(out["nodes"].mean()), 1), "pivots_per_instance": round(pivots / n_local, 1),
                   "pivots_per_s": round(pivots / (kernel_ms * 1e-3)), "rhs_ms": round(float(np.mean([s["rhs_ms"] for s in stats])), 3),
                   "result_gather": (None if gatherer is None else type(gatherer).__name__)},
        "roofline": {"kernel": "k_solve", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": int(upd_bytes), "io_minimum_bytes_per_launch": int(io_bytes),
                     "kernel_ms": round(kernel_ms, 3)},
    }
    if rank == 0:
        # secondary roofline: condensing K1+K2 (SURVEY 8d formula: outputs + inputs), 64 models per launch
        ms = min(model.condense_device(N_t) for _ in range(5))
        nx, ny, nc, nw, nv = d["nx"], d["ny"], d["nc"], d["nomega"], model.nv
        cols = N_t * nv + nx + N_t * nw + 1
        bytes_cond = 8.0 * (N_t * (nc + nx + ny) * cols + (nx + ny + nc) * (nx + nv + nw + 1) + nc * ny)
        ach = args.agents * bytes_cond / (ms * 1e-3) / 1e9
        result["roofline_condense"] = {"kernel": "k_condense_model+k_condense_flat", "bound": "hbm", "achieved": round(ach, 1),
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                       "bytes_per_model": int(bytes_cond), "models": args.agents, "kernel_ms": round(ms, 4)}
        try:        # HBM bytes of the two condensing launches from the same PMC passes (FETCH + WRITE, per launch of each kernel)
            allc = json.load(open(pmc)).get("all", {})
            kb = sum(allc.get(c, {}).get(k, {}).get("kb_per_launch", 0.0) for c in ("FETCH_SIZE", "WRITE_SIZE") for k in ("k_condense_model", "k_condense_flat"))
            result["roofline_condense"]["traffic"] = kb * 1024.0 if kb > 0 else None
        except Exception:
            result["roofline_condense"]["traffic"] = None
        if world == 1 and not args.no_cpu and args.cpu_sample > 0:    # CPU leg: rank 0 at N=1 only
            result["cpu_baseline"] = cpu_baseline(agents, N_p, N_t, x0, om, midx, args.cpu_sample, args.mip_gap, args.node_limit, args.pivot_limit)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
