"""solver trace (opts.reserved bit 0) of one cfg5 instance"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyhybridcontrol_amd import gpu, host, synthetic as syn
i = int(sys.argv[1])
wl = syn.make_workload("cfg5", batch=512); ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=1e-2, max_nodes=100, max_pivots=20000, reserved=1)
out = p.solve(ag["x0"][i:i + 1], ag["omega"][i:i + 1])
print(out["status"], out["obj"], out["nodes"], out["pivots"])
