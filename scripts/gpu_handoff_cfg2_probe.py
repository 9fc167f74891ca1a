"""which cut budget leaves the cfg2 fixture of tests/test_gpu_handoff.py with trees to hand off (and still exactly solvable)?"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from pyhybridcontrol_amd import gpu, host, synthetic as syn
wl = syn.make_workload("cfg2", batch=48); ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
for cr in (0, 1, 2, 3, -1):
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=0.0, max_nodes=100000, cut_rounds=cr)
    ref = p.solve(ag["x0"], ag["omega"])
    out = p.solve_handoff(ag["x0"], ag["omega"], first_nodes=3, sub_nodes=12, rounds=30, max_open=None)
    print("cut_rounds", cr, "ref status", np.unique(ref["status"], return_counts=True), "nodes max", ref["nodes"].max(), "| handoff", {k: v for k, v in out["handoff"].items() if k != "rounds"}, "rounds", len(out["handoff"]["rounds"]), "status", np.unique(out["status"], return_counts=True), flush=True)
    p.close()
