"""ORACLE (test infrastructure): golden OPTIMAL objectives of the benchmarked MILP instances, from a solver nobody here wrote.

    python oracle/gen_solve_golden.py [--scen 32] [--cfg3 64] [--cfg5 32] [--timed 16] [--closed-loop 256] [--procs 8]
    # writes tests/golden/solve_cfg4_bench.npz, solve_cfg3.npz, solve_cfg5.npz, solve_cfg4_timed.npz, solve_cfg4_closed_loop.npz

The reference hands its solve to cvxpy -> Gurobi (controllers/controller_base.py:509); neither is installable here.  The
independent checker is scipy.optimize.milp (HiGHS) at mip_rel_gap = 0 on the ORIGINAL (un-tightened, un-scaled) rows that
`oracle/condense_np.standard_form` -- pinned against the reference's own condensing output -- produces for the seeded
synthetic instances of `bench.make_shard` (cfg4 shard, flattened i = s * 64 + a) and `synthetic.make_workload("cfg3")`.
Per instance: the optimal objective INCLUDING the constant term (what `MpcController.solve` returns,
controller_base.py:533-538), HiGHS's dual bound, its node count and a success flag.  The instances themselves are
regenerated from their seeds by the tests; only these numbers are committed.
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import condense_np as cn  # noqa: E402

_G = {}


def highs_opt(sf, x0, om, gap=0.0, time_limit=120.0):
    from scipy.optimize import Bounds, LinearConstraint, milp
    h = cn.rhs(sf["evo"], x0, om)
    q = cn.lin_cost(sf["cost"], x0, om)
    r = cn.cost_const(sf["cost"]["const_terms"], x0, om)
    res = milp(q, constraints=LinearConstraint(sf["G"], -np.inf, h), bounds=Bounds(sf["lb"], sf["ub"]),
               integrality=sf["is_bin"].astype(int), options=dict(mip_rel_gap=gap, time_limit=time_limit))
    if res.x is None:
        return np.nan, np.nan, -1, 0, None
    return res.fun + r, getattr(res, "mip_dual_bound", np.nan) + r, int(getattr(res, "mip_node_count", -1)), int(res.status == 0), res.x


def _init(kind, n_scen, gap=0.0, time_limit=120.0):
    _G.update(gap=gap, time_limit=time_limit)
    if kind == "cfg4":
        import bench
        agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
        _G.update(agents=agents, N_p=N_p, N_t=N_t, x0=x0, om=om, midx=midx, forms={})
    elif kind == "timed":       # scenario set t = 1 of rank 0 of bench.py's timed region (the generator draws the whole shard at once)
        import bench
        agents, N_p, N_t, _, _, _ = bench.make_shard(64, 1, 0)
        x0, om = bench.step_scenarios(0, 1, 64 * 512)
        n = n_scen * 64
        _G.update(agents=agents, N_p=N_p, N_t=N_t, x0=x0[:n], om=om[:n], midx=np.tile(np.arange(64, dtype=np.int32), n_scen), forms={})
    elif kind == "closed_loop":  # inputs stored in the fixture itself (they are the result of a closed-loop simulation, not of a seed)
        import bench
        agents, N_p, N_t, _, _, _ = bench.make_shard(64, 1, 0)
        z = np.load(os.path.join(ROOT, "tests", "golden", "closed_loop_cfg4_inputs.npz"))
        _G.update(agents=agents, N_p=N_p, N_t=N_t, x0=z["x0"], om=z["omega"], midx=z["model_idx"].astype(np.int32), forms={})
    else:
        from pyhybridcontrol_amd import synthetic as syn
        wl = syn.make_workload(kind, batch=n_scen)
        ag = wl["agents"][0]
        _G.update(agents=[ag], N_p=wl["N_p"], N_t=wl["N_tilde"], x0=ag["x0"], om=ag["omega"],
                  midx=np.zeros(n_scen, np.int32), forms={})


def _form(a):
    if a not in _G["forms"]:
        ag = _G["agents"][a]
        _G["forms"][a] = cn.standard_form(ag["mats"], ag["atoms"], _G["N_p"], _G["N_t"], nu_l=ag["dims"]["nu_l"])
    return _G["forms"][a]


def _one(i):
    a = int(_G["midx"][i])
    t0 = time.perf_counter()
    obj, db, nodes, ok, _ = highs_opt(_form(a), _G["x0"][i], _G["om"][i], gap=_G.get("gap", 0.0), time_limit=_G.get("time_limit", 120.0))
    return i, obj, db, nodes, ok, time.perf_counter() - t0


def run(kind, n_scen, n_inst, procs, out, gap=0.0, time_limit=120.0):
    with mp.Pool(procs, initializer=_init, initargs=(kind, n_scen, gap, time_limit)) as pool:
        obj = np.full(n_inst, np.nan)
        db = np.full(n_inst, np.nan)
        nodes = np.zeros(n_inst, np.int64)
        ok = np.zeros(n_inst, np.uint8)
        secs = np.zeros(n_inst)
        t0 = time.perf_counter()
        for k, (i, o, d, nd, s, t) in enumerate(pool.imap_unordered(_one, range(n_inst), chunksize=1 if kind == "cfg5" else 4)):
            obj[i], db[i], nodes[i], ok[i], secs[i] = o, d, nd, s, t
            if (k + 1) % 128 == 0:
                print("%s: %d/%d  %.0fs" % (kind, k + 1, n_inst, time.perf_counter() - t0), flush=True)
    np.savez_compressed(out, obj=obj, dual_bound=db, nodes=nodes, proven=ok, highs_seconds=secs,
                        n_scen=np.array(n_scen), gap=np.array(gap), solver=np.array("scipy.optimize.milp (HiGHS), mip_rel_gap=%g, original rows" % gap))
    print("wrote", out, "proven", int(ok.sum()), "of", n_inst, "| HiGHS seconds total %.0f max %.1f" % (secs.sum(), secs.max()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scen", type=int, default=16, help="scenarios of the cfg4 shard (x 64 agents)")
    ap.add_argument("--cfg3", type=int, default=64, help="cfg3 instances (agent 0 of make_workload('cfg3'))")
    ap.add_argument("--cfg5", type=int, default=0, help="cfg5 instances (n = 2303, 784 binaries; HiGHS at mip_rel_gap 1e-4 with a 240 s limit each: obj / dual_bound bracket the optimum also when the limit is hit)")
    ap.add_argument("--timed", type=int, default=0, help="scenarios (x 64 agents) of bench.py's timed scenario set t = 1 of rank 0 -> solve_cfg4_timed.npz")
    ap.add_argument("--closed-loop", type=int, default=0, help="instances of tests/golden/closed_loop_cfg4_inputs.npz (made by scripts/cpu_closed_loop.py --dump) -> solve_cfg4_closed_loop.npz")
    ap.add_argument("--cfg5-gap", type=float, default=1e-4)
    ap.add_argument("--cfg5-limit", type=float, default=240.0)
    ap.add_argument("--procs", type=int, default=8)
    args = ap.parse_args()
    gdir = os.path.join(ROOT, "tests", "golden")
    if args.scen > 0:
        run("cfg4", args.scen, args.scen * 64, args.procs, os.path.join(gdir, "solve_cfg4_bench.npz"))
    if args.cfg3 > 0:
        run("cfg3", args.cfg3, args.cfg3, args.procs, os.path.join(gdir, "solve_cfg3.npz"))
    if args.cfg5 > 0:
        run("cfg5", args.cfg5, args.cfg5, args.procs, os.path.join(gdir, "solve_cfg5.npz"), gap=args.cfg5_gap, time_limit=args.cfg5_limit)
    if args.timed > 0:
        run("timed", args.timed, args.timed * 64, args.procs, os.path.join(gdir, "solve_cfg4_timed.npz"))
    if args.closed_loop > 0:
        run("closed_loop", args.closed_loop, args.closed_loop, args.procs, os.path.join(gdir, "solve_cfg4_closed_loop.npz"))
