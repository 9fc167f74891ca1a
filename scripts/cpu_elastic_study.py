"""What implicit soft-constraint slacks would save in the dense dictionary: the oracle counts, per pivot, the penalty
columns whose dictionary column is a unit vector (partner slack basic), those stored once for the pair, and the pivots that merely
exchange a penalty variable with its row's slack.   ORC_ELASTIC_STATS=1 python scripts/cpu_elastic_study.py [n_scen] [threads]"""
import os, sys, ctypes as C, numpy as np
os.environ["ORC_ELASTIC_STATS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
import bench, orc, condense_np as cn, tighten_np
n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 2
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
qs, Gs, hs, sf0 = [], [], [], None
for k in range(x0.shape[0]):
    a = agents[midx[k]]; d = a["dims"]
    sf = a.setdefault("_sf", cn.standard_form(tighten_np.tighten(a["mats"], d, nu_l=d["nu_l"]), a["atoms"], N_p, N_t, nu_l=d["nu_l"]))
    qs.append(cn.lin_cost(sf["cost"], x0[k], om[k])); Gs.append(sf["G"]); hs.append(cn.rhs(sf["evo"], x0[k], om[k])); sf0 = sf
out = orc.solve_milp_batch(qs, Gs, hs, sf0["lb"], sf0["ub"], sf0["is_bin"], threads=threads, gap_rel=1e-2, max_nodes=800, max_pivots=40000, presolve=0)
st = (C.c_double * 5)(); orc.lib().orc_elastic_stats(st)
piv, cheap, implicit, both, pairs = list(st)
n = sf0["G"].shape[1]
print("%d instances, n = %d columns, %.0f penalty columns per instance; %.0f pivots" % (x0.shape[0], n, pairs / max(1, piv), piv))
print("per pivot: %.1f penalty columns have a unit (implicit) column, %.1f share a stored column with their row's slack, %.1f are basic next to a non-basic slack or otherwise stored" % (
    implicit / piv, both / piv, (pairs - implicit - both) / piv))
print("stored columns needed: %.1f of %d (%.1f %%); partner-exchange pivots (one-row updates): %.1f %% of all pivots" % (
    n - implicit / piv, n, 100 * (n - implicit / piv) / n, 100 * cheap / piv))
