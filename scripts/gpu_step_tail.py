"""The slow steps of `bench.py --steps 20`: which instances make a scenario set take 3.3 s instead of 2.2 s.
python scripts/gpu_step_tail.py [set index ...]   (sets as bench.py seeds them: rank 0, set t)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from pyhybridcontrol_amd import gpu, host, synthetic as syn
sets = [int(a) for a in sys.argv[1:]] or [7, 8]
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 512, 0)
d = agents[0]["dims"]
model = gpu.GpuModel([a["mats"] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=800, max_pivots=40000)
prob.upload(x0, om, midx); prob.solve_resident()
for t in sets:
    rng = np.random.Generator(np.random.PCG64([syn.CONFIGS["cfg4"]["seed"], 7919, 0, t]))
    xs, ws = syn.make_scenarios(d["nx"], N_t, x0.shape[0], rng)
    prob.upload(xs, ws, midx); st = prob.solve_resident(); out = prob.download(); tel = prob.telemetry()
    lat = tel["latency_ns"] * 1e-6
    order = np.argsort(-lat)
    print("set %d: kernel %.0f ms; sum of latencies / 256 = %.0f ms; p50 %.1f p99 %.1f max %.0f ms; node-limited %d; instances over 300 ms: %d (%.0f ms of slot time each CU)" % (
        t, st["solve_ms"], lat.sum() / 256, np.median(lat), np.percentile(lat, 99), lat.max(), (out["status"] == 2).sum(), (lat > 300).sum(), lat[lat > 300].sum() / 256))
    for i in order[:8]:
        print("   inst %5d agent %2d: %7.1f ms, status %d, nodes %4d, pivots %6d, %.1f us/pivot" % (i, midx[i], lat[i], out["status"][i], out["nodes"][i], out["pivots"][i], 1e3 * lat[i] / max(1, out["pivots"][i])))
