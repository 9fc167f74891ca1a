"""Tight-gap check of optimality claims: instances the GPU proves optimal at gap 1e-4 against HiGHS at 1e-7 (original rows)."""
import sys, os, numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'oracle'))
import condense_np as cn
from pyhybridcontrol_amd import gpu, synthetic as syn, host
from scipy.optimize import milp, LinearConstraint, Bounds
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 192
kw = {}
for a in sys.argv[2:]:
    k, v = a.split('='); kw[k] = int(v)
wl = syn.make_workload("cfg3", batch=nb)
ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
cost = host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"])
p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, max_nodes=2000, gap_rel=1e-4, **kw)
out = p.solve(ag["x0"], ag["omega"])
raw = cn.standard_form(ag["mats"], ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
bad = []; n_ok = 0
for s in range(nb):
    if out["status"][s] != 0: continue
    h, q = cn.rhs(raw["evo"], ag["x0"][s], ag["omega"][s]), cn.lin_cost(raw["cost"], ag["x0"][s], ag["omega"][s])
    r = cn.cost_const(raw["cost"]["const_terms"], ag["x0"][s], ag["omega"][s])
    ref = milp(q, constraints=LinearConstraint(raw["G"], -np.inf, h), integrality=raw["is_bin"].astype(int), bounds=Bounds(raw["lb"], raw["ub"]), options=dict(mip_rel_gap=1e-7, time_limit=30))
    if ref.status != 0: continue
    n_ok += 1
    rel = (out["obj"][s] - (ref.fun + r)) / max(1.0, abs(ref.fun + r))
    if rel > 2e-4 or rel < -1e-6: bad.append((s, float(out["obj"][s]), float(ref.fun + r), float(rel)))
print('options', kw, 'proven by the GPU and solved by HiGHS:', n_ok, 'claims off by more than 2e-4 (or below the optimum):', bad)
