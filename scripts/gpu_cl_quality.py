"""GPU: the 256 steady-state closed-loop instances (tests/golden/closed_loop_cfg4_inputs.npz) at the bench's options against their HiGHS
optima -- per-instance listing of the worst incumbents (round 3: where does the node-limited tail end up?)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
from pyhybridcontrol_amd import gpu, host
z = np.load(os.path.join(ROOT, "tests", "golden", "closed_loop_cfg4_inputs.npz"))
g = np.load(os.path.join(ROOT, "tests", "golden", "solve_cfg4_closed_loop.npz"))
agents, N_p, N_t, _, _, _ = bench.make_shard(64, 1, 0)
d = agents[0]["dims"]
model = gpu.GpuModel([a["mats"] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
kw = dict(gap_rel=1e-2, max_nodes=800, max_pivots=40000)
kw.update(eval("dict(%s)" % os.environ.get("GPU_KW", "")))
prob = gpu.GpuProblem(model, N_p, N_t, cost, **kw)
out = prob.solve(z["x0"], z["omega"], z["model_idx"].astype(np.int32))
opt = g["obj"]
rel = (out["obj"] - opt) / np.maximum(1e-9, np.abs(opt))
print("proven %.4f  within 1%% %.4f  worst %.4f  nodes %.1f pivots %.1f" % ((out["status"] == 0).mean(), (rel <= 1e-2 + 1e-9).mean(), rel.max(), out["nodes"].mean(), out["pivots"].mean()))
for i in np.argsort(-rel)[:10]:
    print("inst %3d st %d obj %.4f opt %.4f lb %.4f nodes %d pivots %d rel %.3f" % (i, out["status"][i], out["obj"][i], opt[i], out["lower_bound"][i], out["nodes"][i], out["pivots"][i], rel[i]))
