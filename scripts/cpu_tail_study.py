"""CPU study of the bench workload's answer quality: the C oracle (same algorithm as k_solve) on the first instances of the
cfg4 shard against the committed HiGHS optima (tests/golden/solve_cfg4_bench.npz).

    python scripts/cpu_tail_study.py [n_scen=16] [gap=1e-2] [nodes=400] [procs=8]

Prints the status histogram, pivots / nodes per instance, and how far the returned incumbents are from the optimum --
what tests/test_gpu_bench_parity.py asserts on the GPU.  Oracle variants are switched with ORC_* environment variables.
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

_G = {}


def _init(n_scen, gap, nodes):
    import bench
    agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
    _G.update(agents=agents, N_p=N_p, N_t=N_t, x0=x0, om=om, midx=midx, forms={}, gap=gap, nodes=nodes)


def _one(i):
    import condense_np as cn
    import orc
    import tighten_np
    a = int(_G["midx"][i])
    if a not in _G["forms"]:
        ag = _G["agents"][a]
        d = ag["dims"]
        tm = tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"])
        _G["forms"][a] = cn.standard_form(tm, ag["atoms"], _G["N_p"], _G["N_t"], nu_l=d["nu_l"])
    sf = _G["forms"][a]
    x0, om = _G["x0"][i], _G["om"][i]
    h = cn.rhs(sf["evo"], x0, om)
    q = cn.lin_cost(sf["cost"], x0, om)
    rc = cn.cost_const(sf["cost"]["const_terms"], x0, om)
    t0 = time.perf_counter()
    r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], gap_rel=_G["gap"], max_nodes=_G["nodes"], presolve=int(os.environ.get("ORC_PRESOLVE", "4")), max_pivots=40000, **eval("dict(%s)" % os.environ.get("ORC_KW", "")))
    dt = time.perf_counter() - t0
    st = dict(optimal=0, infeasible=1, node_limit=2, numerical=3, unbounded=4)[r["status"]]
    return i, st, r["obj"] + rc, r["lower_bound"] + rc, r["nodes"], r["pivots"], dt


def main():
    n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    gap = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2
    nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    procs = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    gold = np.load(os.path.join(ROOT, "tests", "golden", "solve_cfg4_bench.npz"))
    n = min(n_scen * 64, gold["obj"].size)
    opt = gold["obj"][:n]
    res = np.zeros((n, 6))
    t0 = time.perf_counter()
    with mp.Pool(procs, initializer=_init, initargs=(n_scen, gap, nodes)) as pool:
        for (i, st, obj, lb, nd, pv, dt) in pool.imap_unordered(_one, range(n), chunksize=8):
            res[i] = (st, obj, lb, nd, pv, dt)
    wall = time.perf_counter() - t0
    st, obj, lb, nd, pv, dt = res.T
    scale = np.maximum(1e-9, np.abs(opt))
    rel = (obj - opt) / scale
    print("instances %d  wall %.1fs  cpu %.1fs (%.1f/s per core)" % (n, wall, dt.sum(), n / dt.sum()))
    print("status: optimal %d node_limit %d infeasible %d numerical %d unbounded %d | no incumbent %d" % (
        (st == 0).sum(), (st == 2).sum(), (st == 1).sum(), (st == 3).sum(), (st == 4).sum(), (~np.isfinite(obj)).sum()))
    print("nodes/inst %.1f  pivots/inst %.1f  | node-limited: pivots %.0f nodes %.0f" % (nd.mean(), pv.mean(), pv[st == 2].mean() if (st == 2).any() else 0, nd[st == 2].mean() if (st == 2).any() else 0))
    fin = np.isfinite(obj)
    print("obj below HiGHS optimum by > 1e-6 rel: %d   lower bound above optimum by > 1e-6: %d" % (
        (rel[fin] < -1e-6).sum(), ((lb - opt) / scale > 1e-6).sum()))
    print("OPTIMAL claims outside the gap: %d (worst %.4f)" % (((st == 0) & (rel > gap * (1 + 1e-9) + 1e-9)).sum(), rel[st == 0].max() if (st == 0).any() else 0))
    print("within gap of the optimum: %.3f %%  | > 10 %% above: %d  > 50 %%: %d  worst %.3f" % (
        100.0 * (rel[fin] <= gap + 1e-9).sum() / n, (rel[fin] > 0.1).sum(), (rel[fin] > 0.5).sum(), rel[fin].max()))
    worst = np.argsort(-np.where(fin, rel, np.inf))[:10]
    for i in worst:
        print("  inst %5d st %d obj %.4f opt %.4f lb %.4f rel %.3f nodes %d piv %d" % (i, st[i], obj[i], opt[i], lb[i], rel[i], nd[i], pv[i]))


if __name__ == "__main__":
    main()
