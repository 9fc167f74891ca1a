"""Per-instance presolve (s_presolve, mld_opts.presolve bit 2) against the same solve without it (opts.reserved bit 12) -- DESIGN section 4f:

* the bench shard (n_scen x 64 instances, bench options): kernel time, proven share, nodes / pivots / row updates per instance, and the
  objectives of instances both runs prove must agree within the gap;
* the 256 steady-state closed-loop instances of tests/golden against their HiGHS optima.

    python scripts/gpu_presolve_ab.py [n_scen=512] [node_limit=800]
"""
import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from pyhybridcontrol_amd import gpu, host
n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 800
GAP = 1e-2
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]["dims"]
model = gpu.GpuModel([a["mats"] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
G = os.path.join('/root/repo', 'tests', 'golden')
z = np.load(os.path.join(G, "closed_loop_cfg4_inputs.npz")); gold = np.load(os.path.join(G, "solve_cfg4_closed_loop.npz"))
outs = {}
for r in (4096, 0):
    p = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=GAP, max_nodes=nodes, max_pivots=40000, reserved=r)
    p.upload(x0, om, midx); p.solve_resident(); st = p.solve_resident(); out = p.download(); tel = p.telemetry()
    lat = tel["latency_ns"] * 1e-6
    print("%-9s shard: kernel %.0f ms (%.0f /s), proven %.3f%%, nodes %.1f pivots %.1f, rows updated per pivot %.0f, p50 / p99 latency %.1f / %.1f ms, node-limited %d, other %s" % (
        "presolve" if r == 0 else "without", st["solve_ms"], 1e3 * x0.shape[0] / st["solve_ms"], 100 * (out["status"] == 0).mean(), out["nodes"].mean(), out["pivots"].mean(),
        tel["rows_updated"].sum() / max(1, out["pivots"].sum()), np.percentile(lat, 50), np.percentile(lat, 99), (out["status"] == 2).sum(),
        {int(k): int(v) for k, v in zip(*np.unique(out["status"][(out["status"] != 0) & (out["status"] != 2)], return_counts=True))}), flush=True)
    outs[r] = out
    o2 = p.solve(z["x0"], z["omega"], z["model_idx"].astype(np.int32))
    ok = gold["proven"] == 1
    rel = (o2["obj"][ok] - gold["obj"][ok]) / np.maximum(1.0, np.abs(gold["obj"][ok]))
    print("          steady state: proven %.4f, within gap of the HiGHS optimum %.4f, worst %.4f, below the optimum by more than 1e-6: %d, nodes %.1f pivots %.1f" % (
        (o2["status"] == 0).mean(), (rel <= GAP + 1e-9).mean(), rel.max(), int((rel < -1e-6).sum()), o2["nodes"].mean(), o2["pivots"].mean()), flush=True)
    p.close()
a, b = outs[4096], outs[0]
both = (a["status"] == 0) & (b["status"] == 0)
dif = np.abs(a["obj"][both] - b["obj"][both]) / np.maximum(1.0, np.abs(a["obj"][both]))
print("both proven: %d; objectives differ by more than the gap: %d (largest %.2e); presolve better / worse than without by > 1e-6: %d / %d" % (
    both.sum(), int((dif > GAP).sum()), dif.max(), int((b["obj"][both] < a["obj"][both] - 1e-6).sum()), int((b["obj"][both] > a["obj"][both] + 1e-6).sum())))
lbviol = b["lower_bound"] > np.minimum(a["obj"], b["obj"]) + 1e-6 * np.maximum(1.0, np.abs(b["obj"]))
print("proven bound of the presolve run above the best known objective: %d" % int(lbviol.sum()))
