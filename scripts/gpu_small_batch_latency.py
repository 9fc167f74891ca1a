import os, sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from pyhybridcontrol_amd import gpu, host, synthetic as syn
for name, nb in (("cfg1", 1), ("cfg2", 1), ("cfg2", 256)):
    wl = syn.make_workload(name, batch=nb); ag = wl["agents"][0]; d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]))
    p.upload(ag["x0"], ag["omega"]); p.solve_resident()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); st = p.solve_resident(); ts.append(time.perf_counter() - t0)
    print("%s batch %d: solve_resident wall %.3f ms (min %.3f), device rhs+solve %.3f ms" % (name, nb, 1e3 * np.median(ts), 1e3 * min(ts), st["rhs_ms"] + st["solve_ms"]))
    p.close(); m.close()
