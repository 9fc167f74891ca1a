"""root LP only (no cuts, one node) on the first scenarios of the bench shard: pivots and kernel time with and without the long-step ratio test
(opts.reserved bit 14), beside the oracle's pivots for the same instances.   python scripts/gpu_root_lp.py [n_scen=16]"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np
import bench
from pyhybridcontrol_amd import gpu, host
n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 16
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents])
for res in (16384, 0):
    prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=1, max_pivots=40000, cut_rounds=0, reserved=res)
    prob.upload(x0, om, midx); st = prob.solve_resident(); st = prob.solve_resident()
    out = prob.download()
    print("GPU long-step %s: pivots/inst %.1f  solve_ms %.1f  root bound sum %.6f" % ("off" if res else "on ", st["pivots"] / x0.shape[0], st["solve_ms"], float(np.sum(out["lower_bound"]))), flush=True)
    prob.close()
if "--oracle" in sys.argv:
    import condense_np as cn, orc, tighten_np
    for bf in ("0", "1"):
        os.environ["ORC_BFRT"] = bf
        piv = 0; forms = {}
        for i in range(0, min(256, x0.shape[0])):
            a = int(midx[i])
            if a not in forms:
                dd = agents[a]["dims"]; forms[a] = cn.standard_form(tighten_np.tighten(agents[a]["mats"], dd, nu_l=dd["nu_l"]), agents[a]["atoms"], N_p, N_t, nu_l=dd["nu_l"])
            sf = forms[a]
            r = orc.solve_milp(cn.lin_cost(sf["cost"], x0[i], om[i]), sf["G"], cn.rhs(sf["evo"], x0[i], om[i]), sf["lb"], sf["ub"], sf["is_bin"], gap_rel=1e-2, max_nodes=1, presolve=0, max_pivots=40000, cut_rounds=0)
            piv += r["pivots"]
        print("oracle ORC_BFRT=%s (first %d instances): pivots/inst %.1f" % (bf, min(256, x0.shape[0]), piv / min(256, x0.shape[0])))
