"""Objectives proven (gap 1e-4) under the diagnostic variants of the solver (never-binding rows on/off, Gomory rounds parallel/serial) must agree;
mismatches are checked against HiGHS."""
import sys, os, numpy as np
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, R + '/oracle')
import condense_np as cn, tighten_np
from pyhybridcontrol_amd import gpu, synthetic as syn, host
from scipy.optimize import milp, LinearConstraint, Bounds
wl = syn.make_workload("cfg3", batch=int(sys.argv[1]) if len(sys.argv) > 1 else 96)
ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
cost = host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"])
outs = {}
for flag in (0, 16, 32, 48, 4096, 32768, 4096 + 32768):      # (4096: no per-instance presolve, 32768: rounding cuts one at a time)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, max_nodes=2000, gap_rel=1e-4, reserved=flag)
    outs[flag] = p.solve(ag["x0"], ag["omega"]); p.close()
base = outs[48]
bad = set()
for f, o in outs.items():
    dif = np.abs(o["obj"] - base["obj"]) / np.maximum(1, np.abs(base["obj"]))
    idx = np.where((dif > 2e-4) & (o["status"] == 0) & (base["status"] == 0))[0]
    print('flag', f, 'vs 48: proven', int((o["status"] == 0).sum()), 'mismatching proven', idx.tolist(), [float(dif[i]) for i in idx])
    bad |= set(idx.tolist())
raw = cn.standard_form(ag["mats"], ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
for s in sorted(bad):
    h, q = cn.rhs(raw["evo"], ag["x0"][s], ag["omega"][s]), cn.lin_cost(raw["cost"], ag["x0"][s], ag["omega"][s])
    r = cn.cost_const(raw["cost"]["const_terms"], ag["x0"][s], ag["omega"][s])
    ref = milp(q, constraints=LinearConstraint(raw["G"], -np.inf, h), integrality=raw["is_bin"].astype(int), bounds=Bounds(raw["lb"], raw["ub"]), options=dict(mip_rel_gap=1e-7))
    print('inst', s, 'highs', ref.fun + r, {f: (float(o["obj"][s]), float(o["lower_bound"][s]), int(o["status"][s]), int(o["nodes"][s])) for f, o in outs.items()})
