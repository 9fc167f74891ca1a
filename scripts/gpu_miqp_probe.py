"""MIQP batches (quadratic cost on the state: convex-QP relaxation by simplicial decomposition at every node): throughput and where the time goes
python scripts/gpu_miqp_probe.py [cfg2|cfg3] [batch] [max_nodes]"""
import os, sys, time, ctypes as C, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyhybridcontrol_amd import gpu, host, synthetic as syn, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
max_nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 400
for quad in (False, True):
    wl = syn.make_workload(name, batch=nb, quadratic=quad); ag = wl["agents"][0]; d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=1e-2, max_nodes=max_nodes, max_pivots=40000)
    p.upload(ag["x0"], ag["omega"]); p.solve_resident()
    t0 = time.time(); st = p.solve_resident(); wall = time.time() - t0
    out = p.download(); tel = p.telemetry()
    prof = (C.c_int64 * 8)(); _lib.load().mld_debug_profile(p._h, prof)
    t = np.array(list(prof), dtype=float); tot = tel["latency_ns"].sum() / 10.0      # ticks of 10 ns
    names = ['pivot update', 'simplex selection', 'cuts', 'leaf', 'set_bounds', 'residual/refactor', 'setup', 'master QP (thread 0)']
    print("%s %s batch %d: %.1f ms, %.0f solves/s, nodes %.1f pivots %.1f, optimal %d node-limited %d other %d" % (name, "MIQP" if quad else "MILP", nb, st["solve_ms"], nb / wall,
          out["nodes"].mean(), out["pivots"].mean(), (out["status"] == 0).sum(), (out["status"] == 2).sum(), ((out["status"] != 0) & (out["status"] != 2)).sum()))
    print("   share of in-kernel time: " + ", ".join("%s %.1f%%" % (nm, 100 * v / tot) for nm, v in zip(names, t)))
    p.close(); m.close()
