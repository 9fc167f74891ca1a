"""Bench shard once (second solve: learnt queue order): kernel time, sum of in-kernel latencies / 256, proven share, row updates per pivot.
Used for the A/B of dropping never-binding cut rows below the root (2017 -> 1941 ms, 5666 -> 5125 row-sector updates per pivot)."""
import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from pyhybridcontrol_amd import gpu, host
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 512, 0)
d = agents[0]["dims"]
model = gpu.GpuModel([a["mats"] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
for r in (0,):
    p = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=800, max_pivots=40000, reserved=r)
    p.upload(x0, om, midx); p.solve_resident(); st = p.solve_resident(); out = p.download(); tel = p.telemetry()
    lat = tel["latency_ns"] * 1e-6
    print("reserved %5d: kernel %.0f ms, sum latencies/256 %.0f ms, proven %.3f%%, nodes %.1f pivots %.1f, rows updated per pivot %.0f, node-limited %d" % (
        r, st["solve_ms"], lat.sum() / 256, 100 * (out["status"] == 0).mean(), out["nodes"].mean(), out["pivots"].mean(), tel["rows_updated"].sum() / out["pivots"].sum(), (out["status"] == 2).sum()), flush=True)
    p.close()
