"""MIQP leg of bench.py (cfg3 shape, Q_x = 1e-3 I, 4096 instances): instances that end without an incumbent or unproven, with the per-instance presolve
and without it (opts.reserved bit 12); their statuses, nodes, pivots, and what the other variant returns for them.
    python scripts/gpu_miqp_noinc.py [n_inst=4096]"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from pyhybridcontrol_amd import gpu, host, synthetic as syn
n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wl = syn.make_workload("cfg3", batch=n_inst, quadratic=True)
ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
outs = {}
for r in (0, 4096):
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=1e-2, max_nodes=800, max_pivots=40000, reserved=r)
    p.upload(ag["x0"], ag["omega"]); p.solve_resident(); st = p.solve_resident(); out = p.download(); tel = p.telemetry()
    lat = tel["latency_ns"] * 1e-6
    print("reserved %5d: kernel %.0f ms, statuses %s, no incumbent %d, nodes %.1f pivots %.1f, latency p50 %.1f p99 %.1f max %.1f ms, sum of latencies / 256 = %.0f ms" % (
        r, st["solve_ms"], dict(zip(*np.unique(out["status"], return_counts=True))), int((~np.isfinite(out["obj"])).sum()), out["nodes"].mean(), out["pivots"].mean(),
        np.percentile(lat, 50), np.percentile(lat, 99), lat.max(), lat.sum() / 256), flush=True)
    outs[r] = (out, lat)
    p.close()
a, b = outs[0][0], outs[4096][0]
bad = np.where(~np.isfinite(a["obj"]) | ~np.isfinite(b["obj"]) | (a["status"] > 2) | (b["status"] > 2))[0]
for i in bad[:20]:
    print("inst %4d  presolve: status %d obj %.6g lb %.6g nodes %d pivots %d refac %s %.0f ms | without: status %d obj %.6g lb %.6g nodes %d pivots %d %.0f ms" % (
        i, a["status"][i], a["obj"][i], a["lower_bound"][i], a["nodes"][i], a["pivots"][i], a.get("refactors", np.zeros(n_inst))[i], outs[0][1][i],
        b["status"][i], b["obj"][i], b["lower_bound"][i], b["nodes"][i], b["pivots"][i], outs[4096][1][i]))
both = np.isfinite(a["obj"]) & np.isfinite(b["obj"]) & (a["status"] == 0) & (b["status"] == 0)
dif = np.abs(a["obj"][both] - b["obj"][both]) / np.maximum(1.0, np.abs(b["obj"][both]))
print("both proven %d, largest relative difference %.2e (gap 1e-2)" % (both.sum(), dif.max()))
slow = np.argsort(-outs[0][1])[:8]
print("slowest with presolve:", [(int(i), int(a["status"][i]), int(a["nodes"][i]), int(a["pivots"][i]), round(float(outs[0][1][i])), round(float(outs[4096][1][i]))) for i in slow])
