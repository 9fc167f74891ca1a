"""single cfg5 instances (indices on the command line) under three option sets: default, large pivot limit, no cuts"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyhybridcontrol_amd import gpu, host, synthetic as syn
idx = [int(a) for a in sys.argv[1:]]
wl = syn.make_workload("cfg5", batch=512)
ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=1e-2, max_nodes=100, max_pivots=20000)
out = p.solve(ag["x0"][idx], ag["omega"][idx])
print(out["status"], out["obj"], out["nodes"], out["pivots"], out["stats"]["refactors"], out["stats"]["cuts"])
p2 = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=1e-2, max_nodes=100, max_pivots=200000)
out = p2.solve(ag["x0"][idx], ag["omega"][idx])
print("max_pivots 200000:", out["status"], out["obj"], out["nodes"], out["pivots"])
p3 = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=1e-2, max_nodes=100, max_pivots=200000, cut_rounds=0)
out = p3.solve(ag["x0"][idx], ag["omega"][idx])
print("no cuts:", out["status"], out["obj"], out["nodes"], out["pivots"])
