"""A/B of two oracle builds on the first N cfg4 instances: status counts, pivots, gaps"""
import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import bench
import condense_np as cn, orc, tighten_np
n = int(sys.argv[1]); libs = sys.argv[2:]
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, max(1, (n + 63) // 64), 0)
forms = {}
res = {l: [] for l in libs}
for i in range(n):
    a = int(midx[i]); ag = agents[a]; d = ag["dims"]
    if a not in forms:
        tm = tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"])
        forms[a] = cn.standard_form(tm, ag["atoms"], N_p, N_t, nu_l=d["nu_l"])
    sf = forms[a]
    h = cn.rhs(sf["evo"], x0[i], om[i]); q = cn.lin_cost(sf["cost"], x0[i], om[i])
    for l in libs:
        orc._lib = None; orc._LIB = l
        orc._lib = C.CDLL(l); orc._lib.orc_solve_milp.restype = C.c_int
        r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], gap_rel=1e-2, max_nodes=400, presolve=0, max_pivots=20000)
        res[l].append((r["status"], r["obj"], r["lower_bound"], r["nodes"], r["pivots"]))
for l in libs:
    R = res[l]
    st = [r[0] for r in R]
    piv = np.array([r[4] for r in R]); nodes = np.array([r[3] for r in R])
    lim = [r for r in R if r[0] == "node_limit"]
    gaps = [(r[1] - r[2]) / abs(r[1]) for r in lim if np.isfinite(r[1])]
    print(l, {s: st.count(s) for s in set(st)}, "pivots total", piv.sum(), "mean nodes %.1f" % nodes.mean(),
          "limited: median gap %.4f max gap %.4f" % (np.median(gaps) if gaps else 0, max(gaps) if gaps else 0),
          "pivots in limited %d" % sum(r[4] for r in lim))
a, b = res[libs[0]], res[libs[-1]]
worse = [(i, a[i][1], b[i][1]) for i in range(n) if np.isfinite(a[i][1]) and b[i][1] > a[i][1] * (1 + 1e-2) + 1e-9]
print("instances where the second build's objective is >1% worse:", worse[:10], len(worse))
for l in libs:
    print(l, sorted([(i, round(r[1], 3), round(r[2], 3), r[3], r[4]) for i, r in enumerate(res[l]) if r[0] == "node_limit"], key=lambda t: -(t[1] - t[2]) / abs(t[1]))[:12])
