"""cfg5-size probe (n_h=15, N_p=48): one agent x B scenarios, small node limit; prints status mix and rates"""
import sys, numpy as np
sys.path.insert(0, '.')
from pyhybridcontrol_amd import gpu, host, synthetic as syn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 100
wl = syn.make_workload("cfg5", batch=B)
ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=1e-2, max_nodes=nodes, max_pivots=20000, **(eval('dict(%s)' % sys.argv[3]) if len(sys.argv) > 3 else {}))
p.upload(ag["x0"], ag["omega"]); st = p.solve_resident(); out = p.download(); tel = p.telemetry()
print({k: st[k] for k in ("solve_ms", "pivots", "nodes", "n_optimal", "n_infeasible", "n_node_limit", "n_numerical")})
secs = tel["rows_updated"].sum()
print("n", p.n, "sectors updated", secs, "bytes %.3e" % (secs * 128.0), "GB/s %.1f" % (secs * 128.0 / st["solve_ms"] / 1e6),
      "pivots/s %.0f" % (st["pivots"] / st["solve_ms"] * 1e3), "solves/s %.1f" % (B / st["solve_ms"] * 1e3))
lim = out["status"] == 2
fin = np.isfinite(out["obj"])
print("no incumbent", int((~fin).sum()), "numerical idx", np.where(out["status"] == 3)[0][:5], "infeasible idx", np.where(out["status"] == 1)[0][:8])
if (lim & fin).any():
    g = (out["obj"][lim & fin] - out["lower_bound"][lim & fin]) / np.abs(out["obj"][lim & fin])
    print("gap of limited: median %.4f p90 %.4f max %.4f" % (np.median(g), np.percentile(g, 90), g.max()))
