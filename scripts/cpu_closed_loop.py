"""CPU closed-loop study (round 3): the C oracle in a receding-horizon loop on the first instances of the cfg4 shard -- solve, apply
step 0 to the plant (x+ = A x + [B1 B2 B3] v0 + B4 w0 + b5, the arithmetic of mld_advance_batch), rotate the forecast -- with and
without the shifted previous plan as MIP start.

    python scripts/cpu_closed_loop.py <tag> [n_inst=512] [steps=24] [warm=0|1] [gap=1e-2] [nodes=800] [procs=8]
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
    _G.update(agents=agents, N_p=N_p, N_t=N_t, midx=midx, forms={}, gap=gap, nodes=nodes)


def _form(a):
    import condense_np as cn
    import tighten_np
    if a not in _G["forms"]:
        ag = _G["agents"][a]
        d = ag["dims"]
        tm = tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"])
        _G["forms"][a] = cn.standard_form(tm, ag["atoms"], _G["N_p"], _G["N_t"], nu_l=d["nu_l"])
    return _G["forms"][a]


def _one(job):
    import condense_np as cn
    import orc
    i, x0, om, start = job
    sf = _form(int(_G["midx"][i]))
    h = cn.rhs(sf["evo"], x0, om)
    q = cn.lin_cost(sf["cost"], x0, om)
    rc = cn.cost_const(sf["cost"]["const_terms"], x0, om)
    r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], x_start=start, gap_rel=_G["gap"], max_nodes=_G["nodes"], presolve=0,
                       max_pivots=40000, **eval("dict(%s)" % os.environ.get("ORC_KW", "")))
    st = dict(optimal=0, infeasible=1, node_limit=2, numerical=3, unbounded=4)[r["status"]]
    return i, st, r["obj"] + rc, r["lower_bound"] + rc, r["nodes"], r["pivots"], r["work"], r["x"]


def main():
    tag = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    warm = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    gap = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-2
    nodes = int(sys.argv[6]) if len(sys.argv) > 6 else 800
    procs = int(sys.argv[7]) if len(sys.argv) > 7 else 8
    import bench
    n_scen = (n + 63) // 64
    agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
    x0, om = x0[:n].copy(), om[:n].copy()
    d = agents[0]["dims"]
    nv = d["nu"] + d["ndelta"] + d["nz"] + d["nmu"]
    nw = d["nomega"]
    plan = None
    hist = []
    with mp.Pool(procs, initializer=_init, initargs=(n_scen, gap, nodes)) as pool:
        for k in range(steps):
            jobs = [(i, x0[i], om[i], (plan[i] if (warm and plan is not None and plan[i] is not None) else None)) for i in range(n)]
            t0 = time.perf_counter()
            out = pool.map(_one, jobs, chunksize=4)
            wall = time.perf_counter() - t0
            st = np.array([o[1] for o in out]); obj = np.array([o[2] for o in out]); lb = np.array([o[3] for o in out])
            nd = np.array([o[4] for o in out]); pv = np.array([o[5] for o in out]); wk = np.array([o[6] for o in out])
            gapr = (obj - lb) / np.maximum(1e-9, np.abs(obj))
            print("[%s] step %2d: proven %.2f%% node-limited %d | nodes %.1f pivots %.1f work %.0f | p99 piv %.0f max gap of limited %.3f | wall %.1fs" % (
                tag, k, 100.0 * (st == 0).mean(), (st == 2).sum(), nd.mean(), pv.mean(), wk.mean(), np.percentile(pv, 99),
                gapr[st == 2].max() if (st == 2).any() else 0.0, wall), flush=True)
            hist.append((100.0 * (st == 0).mean(), nd.mean(), pv.mean(), wk.mean()))
            # plant update + forecast rotation + shifted plan
            newplan = []
            for i in range(n):
                v = out[i][7]
                ag = agents[int(midx[i])]
                M = ag["mats"]
                if v is None:
                    newplan.append(None)
                    continue
                V = v.reshape(N_t, nv)
                W = om[i].reshape(N_t, nw)
                Bv = np.hstack([M.get(kk) if M.get(kk) is not None and np.size(M.get(kk)) else np.zeros((d["nx"], dd)) for kk, dd in (("B1", d["nu"]), ("B2", d["ndelta"]), ("B3", d["nz"]))])
                Bv = np.hstack([Bv, np.zeros((d["nx"], d["nmu"]))])
                x0[i] = M["A"] @ x0[i] + Bv @ V[0] + M["B4"] @ W[0] + M["b5"].ravel()
                om[i] = np.roll(W, -1, axis=0).ravel()
                newplan.append(np.vstack([V[1:], V[-1:]]).ravel())
            plan = newplan
    if os.environ.get("CL_DUMP"):      # steady-state inputs as a fixture: tests/golden/closed_loop_cfg4_inputs.npz
        np.savez_compressed(os.environ["CL_DUMP"], x0=x0, omega=om, model_idx=midx[:n].astype(np.int16), steps=np.array(steps),
                            note=np.array("inputs of the first %d instances of the cfg4 shard after %d closed-loop MPC steps (oracle at MIPGap %g / NodeLimit %d, "
                                          "plant = control model, forecast rotated): scripts/cpu_closed_loop.py" % (n, steps, gap, nodes)))
    h = np.array(hist)
    print("[%s] mean over steps: proven %.2f%% nodes %.1f pivots %.1f work %.0f | last 8 steps: proven %.2f%% pivots %.1f work %.0f" % (
        tag, h[:, 0].mean(), h[:, 1].mean(), h[:, 2].mean(), h[:, 3].mean(), h[-8:, 0].mean(), h[-8:, 2].mean(), h[-8:, 3].mean()))


if __name__ == "__main__":
    main()
