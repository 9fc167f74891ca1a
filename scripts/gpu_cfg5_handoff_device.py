"""cfg5 (128 golden instances) with the in-kernel hand-off: proven share and time over a few settings
   python scripts/gpu_cfg5_handoff_device.py "400,400,4,64,160;200,200,8,64,160" """
import sys, time, os
sys.path.insert(0, '.')
import numpy as np
from pyhybridcontrol_amd import gpu, host, synthetic as syn
gold = np.load("tests/golden/solve_cfg5.npz")
nb = int(gold["n_scen"])
wl = syn.make_workload("cfg5", batch=nb); ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=1e-2, max_nodes=400, max_pivots=160000)
ok = np.isfinite(gold["obj"]) & np.isfinite(gold["dual_bound"])
def report(tag, out, wall):
    rel = (out["obj"][ok] - gold["obj"][ok]) / np.maximum(1e-9, np.abs(gold["obj"][ok]))
    print("%s: %.1f s  proven %d of %d  within 1%% of HiGHS %d of %d  worst %.3f  below dual bound %d  %s" % (tag, wall, (out["status"] == 0).sum(), nb, (rel <= 1e-2).sum(), ok.sum(), rel.max(),
          int((out["obj"][ok] < gold["dual_bound"][ok] - 1e-6 * np.abs(gold["obj"][ok])).sum()), out.get("handoff", "")), flush=True)
t0 = time.time(); out = p.solve(ag["x0"], ag["omega"]); report("plain 400 nodes", out, time.time() - t0)
for c in (sys.argv[1] if len(sys.argv) > 1 else "400,400,4,64,160").split(";"):
    fn, sn, mg, mc, mt = [int(v) for v in c.split(",")]
    t0 = time.time()
    out = p.solve_handoff_device(ag["x0"], ag["omega"], first_nodes=fn, sub_nodes=sn, max_gen=mg, max_children=mc, max_tree=mt, room_factor=64.0)
    report("hand-off first %d sub %d gen %d children %d tree %d" % (fn, sn, mg, mc, mt), out, time.time() - t0)
