"""GPU: the cfg5 golden instances at the test's options; lists instances that end without an incumbent"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from pyhybridcontrol_amd import gpu, host, synthetic as syn
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
wl = syn.make_workload("cfg5", batch=nb)
ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
kw = dict(gap_rel=1e-2, max_nodes=400, max_pivots=160000)
kw.update(eval("dict(%s)" % os.environ.get("GPU_KW", "")))
p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), **kw)
out = p.solve(ag["x0"], ag["omega"])
tel = p.telemetry()
print(out["stats"])
bad = np.where(~np.isfinite(out["obj"]))[0]
print("no incumbent:", bad)
for i in bad:
    print(i, "status", out["status"][i], "nodes", out["nodes"][i], "pivots", out["pivots"][i], "lb", out["lower_bound"][i], "ms", tel["latency_ns"][i] * 1e-6)
print("proven", (out["status"] == 0).sum(), "of", nb, "max pivots", out["pivots"].max(), "max ms", tel["latency_ns"].max() * 1e-6)
