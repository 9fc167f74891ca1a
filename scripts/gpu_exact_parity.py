import os, sys, numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'oracle'))
import condense_np as cn, orc, tighten_np
from pyhybridcontrol_amd import gpu, host, synthetic as syn
for name, nb in (("cfg2", 48), ("cfg1", 16)):
    wl = syn.make_workload(name, batch=nb); ag = wl["agents"][0]; d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), max_nodes=20000)
    out = p.solve(ag["x0"], ag["omega"])
    tm = tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"]); sf = cn.standard_form(tm, ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    both = 0; worst = 0.0; mism = []
    for s in range(nb):
        h = cn.rhs(sf["evo"], ag["x0"][s], ag["omega"][s]); q = cn.lin_cost(sf["cost"], ag["x0"][s], ag["omega"][s])
        r0 = cn.cost_const(sf["cost"]["const_terms"], ag["x0"][s], ag["omega"][s])
        ref = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], max_nodes=20000, presolve=4)
        if ref["status"] == "optimal" and out["status"][s] == 0:
            both += 1; e = abs(out["obj"][s] - ref["obj"] - r0) / max(1.0, abs(ref["obj"] + r0)); worst = max(worst, e)
            if e > 1e-6: mism.append((s, out["obj"][s], ref["obj"] + r0))
    print(name, "instances", nb, "both proven optimal (gap 1e-9)", both, "worst relative objective difference %.2e" % worst, "mismatches", mism[:3])
