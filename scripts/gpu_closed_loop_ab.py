"""GPU: the closed loop of bench.py (mld_advance_batch between solves) for N steps, cold against the MIP start from the shifted previous plan
(mld_warm_start_from_previous): agent-solves/s, proven share, nodes and pivots per instance, step by step.

    python scripts/gpu_closed_loop_ab.py [steps=24] [n_scen=512] [modes=cold,warm]
"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
from pyhybridcontrol_amd import gpu, host
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n_scen = int(sys.argv[2]) if len(sys.argv) > 2 else 512
modes = (sys.argv[3] if len(sys.argv) > 3 else "cold,warm").split(",")
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]["dims"]
model = gpu.GpuModel([a["mats"] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
kw = dict(gap_rel=1e-2, max_nodes=800, max_pivots=40000)
kw.update(eval("dict(%s)" % os.environ.get("GPU_KW", "")))
for mode in modes:
    prob = gpu.GpuProblem(model, N_p, N_t, cost, **kw)
    prob.upload(x0, om, midx)
    for k in range(steps):
        if k:
            prob.advance()
            if mode == "warm":
                prob.warm_start_from_previous(1)
        t0 = time.perf_counter()
        st = prob.solve_resident()
        dt = time.perf_counter() - t0
        n = x0.shape[0]
        print("[%s] step %2d: %8.1f solves/s  proven %.4f  nodes %.1f pivots %.1f  kernel %.0f ms" % (mode, k, n / dt, st["n_optimal"] / n, st["nodes"] / n, st["pivots"] / n, st["solve_ms"]), flush=True)
    prob.close()
