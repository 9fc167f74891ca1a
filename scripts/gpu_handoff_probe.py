"""GPU: the exact-contract leg of bench.py (gap 1e-6 on the cfg4 shard) with sub-tree hand-off (GpuProblem.solve_handoff) for several
(first_nodes, sub_nodes, rounds) settings: wall time, proven share, objectives against the committed HiGHS optima.

    python scripts/gpu_handoff_probe.py [n_scen=512] ["500,2000,4;1000,2000,4"]
"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
from pyhybridcontrol_amd import gpu, host
n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 512
combos = [tuple(int(x) for x in c.split(",")) for c in (sys.argv[2] if len(sys.argv) > 2 else "500,2000,4;1000,2000,4;2000,4000,3").split(";")]
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]["dims"]
model = gpu.GpuModel([a["mats"] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
gold = np.load(os.path.join(ROOT, "tests", "golden", "solve_cfg4_bench.npz"))["obj"]
kw = dict(gap_rel=1e-6, max_nodes=20000, max_pivots=400000)
kw.update(eval("dict(%s)" % os.environ.get("GPU_KW", "")))
prob = gpu.GpuProblem(model, N_p, N_t, cost, **kw)
n = x0.shape[0]
for fn, sn, rd in combos:
    t0 = time.perf_counter()
    out = prob.solve_handoff(x0, om, midx, first_nodes=fn, sub_nodes=sn, rounds=rd, sub_opts=eval("dict(%s)" % os.environ.get("SUB_KW", "")) or None, max_open=(int(os.environ["MAX_OPEN"]) if "MAX_OPEN" in os.environ else None))
    wall = time.perf_counter() - t0
    k = min(n, gold.size)
    rel = (out["obj"][:k] - gold[:k]) / np.maximum(1e-9, np.abs(gold[:k]))
    pk = out["status"][:k] == 0
    h = out["handoff"]
    print("first %5d sub %5d rounds %d: %.2f s = %.0f solves/s | proven %.5f | handed off %d unfinished %d | rounds %s | worst proven diff %.2e worst above %.4f below %d | pivots/inst %.0f" % (
        fn, sn, rd, wall, n / wall, (out["status"] == 0).mean(), h["handed_off"], h["unfinished"],
        [(r["sub_instances"], round(r["ms"]), r.get("parents_left")) for r in h["rounds"]], np.abs(rel[pk]).max(), rel.max(), (rel < -1e-6).sum(), out["pivots"].mean()), flush=True)
