import sys, os, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import bench
import condense_np as cn, orc, tighten_np
i = int(sys.argv[1])
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, max(8, i // 64 + 1), 0)
a = int(midx[i]); ag = agents[a]; d = ag["dims"]
tm = tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"])
sf = cn.standard_form(tm, ag["atoms"], N_p, N_t, nu_l=d["nu_l"])
h = cn.rhs(sf["evo"], x0[i], om[i]); q = cn.lin_cost(sf["cost"], x0[i], om[i])
np.savez("/tmp/inst_%d.npz" % i, q=q, G=sf["G"], h=h, lb=sf["lb"], ub=sf["ub"], is_bin=sf["is_bin"])
r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], gap_rel=1e-2, max_nodes=int(sys.argv[2]) if len(sys.argv)>2 else 400, presolve=0, max_pivots=20000)
print({k: v for k, v in r.items() if k != 'x'})
