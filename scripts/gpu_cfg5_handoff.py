"""GPU: the cfg5 golden instances (n = 2303, 784 binaries) with sub-tree hand-off against HiGHS's brackets"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from pyhybridcontrol_amd import gpu, host, synthetic as syn
gold = np.load(os.path.join(ROOT, "tests", "golden", "solve_cfg5.npz"))
nb = int(gold["n_scen"])
wl = syn.make_workload("cfg5", batch=nb)
ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=1e-2, max_nodes=400, max_pivots=160000)
for fn, sn, rd, mo in ((400, 400, 4, 64), (400, 800, 6, 128)):
    t0 = time.perf_counter()
    out = p.solve_handoff(ag["x0"], ag["omega"], first_nodes=fn, sub_nodes=sn, rounds=rd, max_open=mo)
    wall = time.perf_counter() - t0
    ok = np.isfinite(gold["obj"]) & np.isfinite(gold["dual_bound"])
    rel = (out["obj"][ok] - gold["obj"][ok]) / np.abs(gold["obj"][ok])
    print("first %d sub %d rounds %d max_open %d: %.1f s | proven %d of %d | within 1%% of HiGHS %d of %d worst %.3f | lb above HiGHS incumbent: %d | obj below HiGHS bound: %d | %s" % (
        fn, sn, rd, mo, wall, (out["status"] == 0).sum(), nb, (rel <= 1e-2).sum(), ok.sum(), rel.max(),
        (out["lower_bound"][ok] > gold["obj"][ok] * (1 + 1e-6) + 1e-9).sum(), (out["obj"][ok] < gold["dual_bound"][ok] * (1 - 1e-6) - 1e-9).sum(),
        [(r["sub_instances"], round(r["ms"]), r.get("parents_left")) for r in out["handoff"]["rounds"]]), flush=True)
