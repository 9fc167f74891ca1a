"""CPU study of the node-limited tail: oracle vs HiGHS on the first instances of the cfg4 shard."""
import sys, os, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import bench
import condense_np as cn, orc, tighten_np
from scipy.optimize import milp, LinearConstraint, Bounds
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 8, 0)
forms = {}
rows = []
for i in range(n):
    a = int(midx[i]); ag = agents[a]; d = ag["dims"]
    if a not in forms:
        tm = tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"])
        forms[a] = cn.standard_form(tm, ag["atoms"], N_p, N_t, nu_l=d["nu_l"])
    sf = forms[a]
    h = cn.rhs(sf["evo"], x0[i], om[i]); q = cn.lin_cost(sf["cost"], x0[i], om[i])
    t = time.perf_counter()
    r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], gap_rel=1e-2, max_nodes=400, presolve=0, max_pivots=20000)
    t1 = time.perf_counter() - t
    t = time.perf_counter()
    res = milp(q, constraints=LinearConstraint(sf["G"], -np.inf, h), bounds=Bounds(sf["lb"], sf["ub"]), integrality=sf["is_bin"].astype(int), options=dict(mip_rel_gap=1e-2))
    t2 = time.perf_counter() - t
    rows.append((i, r["status"], r["obj"], r["lower_bound"], r["root_lp"], r["root_bound"], r["nodes"], r["pivots"], t1, res.fun, getattr(res, "mip_dual_bound", np.nan), getattr(res, "mip_node_count", -1), t2))
    print("%3d %-10s obj %.6f lb %.6f rootlp %.6f rootcut %.6f nodes %4d piv %5d %.3fs | highs %.6f db %.6f nodes %s %.3fs" % rows[-1], flush=True)
