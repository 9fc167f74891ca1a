"""CPU study harness (round 3): the C oracle on the first n instances of the cfg4 shard against the committed HiGHS optima, with the
per-instance results saved so that variants (ORC_* environment switches, ORC_KW option overrides) can be compared instance by instance.

    python scripts/cpu_study.py <tag> [n_inst=2048] [gap=1e-6] [nodes=20000] [pivots=40000] [procs=8]
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


def _init(n_scen, gap, nodes, pivots):
    import bench
    agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
    _G.update(agents=agents, N_p=N_p, N_t=N_t, x0=x0, om=om, midx=midx, forms={}, gap=gap, nodes=nodes, pivots=pivots)


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
    r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], gap_rel=_G["gap"], max_nodes=_G["nodes"], presolve=int(os.environ.get("ORC_PRESOLVE", "4")),
                       max_pivots=_G["pivots"], **eval("dict(%s)" % os.environ.get("ORC_KW", "")))
    dt = time.perf_counter() - t0
    st = dict(optimal=0, infeasible=1, node_limit=2, numerical=3, unbounded=4)[r["status"]]
    return i, st, r["obj"] + rc, r["lower_bound"] + rc, r["nodes"], r["pivots"], dt, r["root_bound"] + rc, r["work"], r["cuts"], r["bland"], r["rebuilds"], *r["phase_work"], r["root_lp"] + rc


def main():
    tag = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    gap = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
    nodes = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
    pivots = int(sys.argv[5]) if len(sys.argv) > 5 else 40000
    procs = int(sys.argv[6]) if len(sys.argv) > 6 else 8
    gold = np.load(os.path.join(ROOT, "tests", "golden", "solve_cfg4_bench.npz"))
    n = min(n, gold["obj"].size)
    n_scen = (n + 63) // 64
    opt = gold["obj"][:n]
    res = np.zeros((n, 18))
    t0 = time.perf_counter()
    with mp.Pool(procs, initializer=_init, initargs=(n_scen, gap, nodes, pivots)) as pool:
        for (i, st, obj, lb, nd, pv, dt, rb, wk, ct, bl, rbd, *pw) in pool.imap_unordered(_one, range(n), chunksize=4):
            res[i] = (st, obj, lb, nd, pv, dt, rb, wk, ct, bl, rbd, *pw)
    wall = time.perf_counter() - t0
    st, obj, lb, nd, pv, dt, rb, wk, ct, bl, rbd = res.T[:11]
    pw = res[:, 11:17]
    rlp = res[:, 17]
    os.makedirs("/tmp/study", exist_ok=True)
    np.savez("/tmp/study/%s.npz" % tag, st=st, obj=obj, lb=lb, nd=nd, pv=pv, dt=dt, rb=rb, opt=opt, wk=wk, ct=ct, bl=bl, rbd=rbd, pw=pw, rlp=rlp)
    scale = np.maximum(1e-9, np.abs(opt))
    rel = (obj - opt) / scale
    fin = np.isfinite(obj)
    print("[%s] instances %d gap %g nodes %d  wall %.1fs  cpu %.1fs (%.2f/s per core)" % (tag, n, gap, nodes, wall, dt.sum(), n / dt.sum()))
    print("status: optimal %d node_limit %d infeasible %d numerical %d unbounded %d | no incumbent %d" % (
        (st == 0).sum(), (st == 2).sum(), (st == 1).sum(), (st == 3).sum(), (st == 4).sum(), (~fin).sum()))
    print("nodes/inst %.1f  pivots/inst %.1f  | median piv %.0f p90 %.0f p99 %.0f max %.0f" % (nd.mean(), pv.mean(), np.median(pv), np.percentile(pv, 90), np.percentile(pv, 99), pv.max()))
    print("work (row updates)/inst %.0f  rows/pivot %.1f  cuts/inst %.1f" % (wk.mean(), wk.sum() / max(1.0, pv.sum()), ct.mean()))
    print("pivots under Bland's rule: %.1f%% of all  | instances with a root rebuild: %d  | pivots of those instances: %.1f%% of all" % (100 * bl.sum() / pv.sum(), (rbd > 0).sum(), 100 * pv[rbd > 0].sum() / pv.sum()))
    print("plain root LP gap (opt-rootlp)/opt: <=1e-6 %.1f%%  <=1e-3 %.1f%%  <=1e-2 %.1f%%  <=5e-2 %.1f%%" % tuple(100 * ((opt - rlp) / np.maximum(1e-9, np.abs(opt)) <= x).mean() for x in (1e-6, 1e-3, 1e-2, 5e-2)))
    print("work split: root LP %.1f%%  cuts %.1f%%  IDS %.1f%%  DIVE %.1f%%  RINS %.1f%%  FINAL %.1f%%" % tuple(100 * pw.sum(0) / max(1.0, wk.sum())))
    o = np.sort(pv)[::-1]
    print("share of pivots: top 0.5%% %.1f%%  top 1%% %.1f%%  top 5%% %.1f%%  top 10%% %.1f%%" % tuple(100 * o[:max(1, int(n * f))].sum() / o.sum() for f in (0.005, 0.01, 0.05, 0.1)))
    print("obj below optimum by > 1e-6 rel: %d   lower bound above optimum by > 1e-6: %d   OPTIMAL outside gap: %d" % (
        (rel[fin] < -1e-6).sum(), ((lb - opt) / scale > 1e-6).sum(), ((st == 0) & (rel > gap * (1 + 1e-9) + 1e-9)).sum()))
    print("within 1e-6: %.3f %%  within 1%%: %.3f %%  worst rel above %.4f  | root gap (opt-rootbound)/opt: median %.4f p90 %.4f max %.4f" % (
        100.0 * (rel[fin] <= 1e-6).sum() / n, 100.0 * (rel[fin] <= 1e-2).sum() / n, rel[fin].max(),
        np.median((opt - rb) / scale), np.percentile((opt - rb) / scale, 90), ((opt - rb) / scale).max()))
    worst = np.argsort(-pv)[:12]
    for i in worst:
        print("  inst %5d st %d obj %.4f opt %.4f lb %.4f rootb %.4f rel %.5f nodes %d piv %d" % (i, st[i], obj[i], opt[i], lb[i], rb[i], rel[i], nd[i], pv[i]))


if __name__ == "__main__":
    main()
