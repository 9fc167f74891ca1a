"""Bench shard (second solve) and the steady-state fixture under solver option sets:
    python scripts/gpu_opts_ab.py [n_scen=512] "mir_per_round=10" "mir_per_round=10 cuts_per_round=40" ...   (the empty string = defaults)"""
import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from pyhybridcontrol_amd import gpu, host
n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sets = sys.argv[2:] or [""]
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]["dims"]
model = gpu.GpuModel([a["mats"] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
G = os.path.join('/root/repo', 'tests', 'golden')
z = np.load(os.path.join(G, "closed_loop_cfg4_inputs.npz")); gold = np.load(os.path.join(G, "solve_cfg4_closed_loop.npz"))
x1, o1 = bench.step_scenarios(0, 19, 64 * n_scen)
import datetime as dt
from pyhybridcontrol_amd import synthetic as syn
t19 = dt.datetime(2018, 12, 10, 5, 0) + dt.timedelta(seconds=syn.TS * 19)
cost19 = host.stack_costs([host.cost_from_atoms(syn.make_cost(syn.CONFIGS["cfg4"]["n_h"], N_t, a["params"], t0=t19), d, N_p, N_t) for a in agents])
for sset in sets:
    kw = {}
    for kv in sset.split():
        k, v = kv.split("="); kw[k] = float(v) if ("." in v or "e" in v) else int(v)
    opts = dict(gap_rel=1e-2, max_nodes=800, max_pivots=40000); opts.update(kw)
    p = gpu.GpuProblem(model, N_p, N_t, cost, **opts)
    p.upload(x0, om, midx); p.solve_resident(); st = p.solve_resident(); out = p.download()
    line = "%-44s 05:00 tariff: %.0f ms (%.0f /s) proven %.3f%% nodes %.1f pivots %.1f cuts %.1f" % (sset or "(defaults)", st["solve_ms"], 1e3 * x0.shape[0] / st["solve_ms"], 100 * (out["status"] == 0).mean(), out["nodes"].mean(), out["pivots"].mean(), out["cuts"].mean() if "cuts" in out else -1)
    p.set_cost(cost19); p.upload(x1, o1, midx); p.solve_resident(); st = p.solve_resident(); out = p.download()
    line += " | 09:45 tariff: %.0f ms (%.0f /s) proven %.3f%% pivots %.1f" % (st["solve_ms"], 1e3 * x1.shape[0] / st["solve_ms"], 100 * (out["status"] == 0).mean(), out["pivots"].mean())
    p.set_cost(cost)
    o2 = p.solve(z["x0"], z["omega"], z["model_idx"].astype(np.int32))
    ok = gold["proven"] == 1
    rel = (o2["obj"][ok] - gold["obj"][ok]) / np.maximum(1.0, np.abs(gold["obj"][ok]))
    line += " | steady state: proven %.4f within gap %.4f worst %.4f" % ((o2["status"] == 0).mean(), (rel <= 1e-2 + 1e-9).mean(), rel.max())
    print(line, flush=True)
    p.close()
