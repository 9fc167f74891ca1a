"""closed loop to step N (eager MIP start), then the anatomy of the last step: share of the step's time spent on instances that end unproven, their reported gaps,
nodes and pivots.   python scripts/gpu_cl_tail.py [steps=24] [reserved=131072] [node_limit=800]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import bench
from pyhybridcontrol_amd import gpu, host
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reserved = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 800
ho = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else None      # in-kernel hand-off all along: sub_nodes,max_gen,max_children,max_tree
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 512, 0)
d = agents[0]["dims"]
model = gpu.GpuModel([a["mats"] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=nl, max_pivots=40000, reserved=reserved)
if ho:
    prob.set_handoff(True, sub_nodes=ho[0], max_gen=ho[1], max_children=ho[2], max_tree=ho[3], room_factor=2.0)
prob.upload(x0, om, midx)
for k in range(steps):
    if k:
        prob.advance(); prob.warm_start_from_previous(1)
    st = prob.solve_resident()
out, tel = prob.download(), prob.telemetry()
print("node limit %d hand-off %s: step %d %.0f solves/s proven %.4f %s" % (nl, ho, steps - 1, x0.shape[0] / (st["solve_ms"] * 1e-3), st["n_optimal"] / x0.shape[0], prob.handoff_stats() if ho else ""))
lat = tel["latency_ns"] * 1e-6
lim = out["status"] == 2
with np.errstate(invalid="ignore", divide="ignore"):
    gap = (out["obj"] - out["lower_bound"]) / np.maximum(1e-9, np.abs(out["obj"]))
n = lat.size
print("step %d: kernel %.0f ms; sum of in-kernel latencies / 256 = %.0f ms; unproven %d (%.2f %%) take %.0f ms / 256 = %.1f %% of the busy time" % (
    steps - 1, st["solve_ms"], lat.sum() / 256, lim.sum(), 100 * lim.mean(), lat[lim].sum() / 256, 100 * lat[lim].sum() / lat.sum()))
print("unproven: gap quantiles 10/50/90/max %s ; nodes mean %.0f ; pivots mean %.0f ; latency mean %.0f ms" % (
    np.round(np.nanquantile(gap[lim], [0.1, 0.5, 0.9, 1.0]), 4), out["nodes"][lim].mean(), out["pivots"][lim].mean(), lat[lim].mean()))
for thr in (0.015, 0.02, 0.03, 0.05):
    print("   gap <= %.3f: %d of %d" % (thr, int((gap[lim] <= thr).sum()), int(lim.sum())))
pr = ~lim
print("proven: nodes mean %.1f pivots mean %.0f latency mean %.1f ms p50 %.1f p99 %.1f ; solved at the root (nodes <= 2): %.1f %%" % (
    out["nodes"][pr].mean(), out["pivots"][pr].mean(), lat[pr].mean(), np.median(lat[pr]), np.quantile(lat[pr], 0.99), 100 * (out["nodes"][pr] <= 2).mean()))
xk, wk = prob.inputs()
np.savez("gpurun_out/cl_step_inputs.npz", x0=xk, omega=wk, midx=midx, status=out["status"], obj=out["obj"], lb=out["lower_bound"], nodes=out["nodes"], pivots=out["pivots"], lat=lat)
