"""cfg5 shape (n = 2303, 784 binaries): per instance, where the gap sits -- incumbent against the HiGHS optimum, proven bound against it:
python scripts/gpu_cfg5_gap_study.py [max_nodes] [gap_rel] [key=value solver options ...]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyhybridcontrol_amd import gpu, host, synthetic as syn
gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'solve_cfg5.npz'))
nb = int(gold["n_scen"])
max_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 400
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2
kw = {}
for a in sys.argv[3:]:
    k, v = a.split("="); kw[k] = int(v)
wl = syn.make_workload("cfg5", batch=nb); ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=gap, max_nodes=max_nodes, max_pivots=40000 * max(1, max_nodes // 400), **kw)
t0 = time.time(); out = p.solve(ag["x0"], ag["omega"]); wall = time.time() - t0
hi, lo = gold["obj"], gold["dual_bound"]
sc = np.maximum(1e-9, np.abs(hi))
print("cfg5 %d instances, NodeLimit %d, gap %g, %s: %.2f s; proven %d, node-limited %d" % (nb, max_nodes, gap, kw, wall, (out["status"] == 0).sum(), (out["status"] == 2).sum()))
print("inst status  nodes pivots   incumbent-vs-HiGHS   bound-vs-HiGHS   own gap")
for i in range(nb):
    print("%3d   %d    %5d %6d   %+9.4f%%          %+9.4f%%      %8.4f%%" % (i, out["status"][i], out["nodes"][i], out["pivots"][i], 100 * (out["obj"][i] - hi[i]) / sc[i],
                                                                   100 * (out["lower_bound"][i] - hi[i]) / sc[i], 100 * (out["obj"][i] - out["lower_bound"][i]) / max(1e-9, abs(out["obj"][i]))))
print("median incumbent excess %.4f%%, median bound deficit %.4f%%" % (100 * np.median((out["obj"] - hi) / sc), 100 * np.median((hi - out["lower_bound"]) / sc)))
