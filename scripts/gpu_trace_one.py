"""GPU: solver trace of ONE steady-state closed-loop instance (needs a -DMLD_TRACE build: MLD_CXXFLAGS=-DMLD_TRACE python -m pyhybridcontrol_amd.build --force)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
from pyhybridcontrol_amd import gpu, host
i = int(sys.argv[1])
z = np.load(os.path.join(ROOT, "tests", "golden", "closed_loop_cfg4_inputs.npz"))
agents, N_p, N_t, _, _, _ = bench.make_shard(64, 1, 0)
d = agents[0]["dims"]
model = gpu.GpuModel([a["mats"] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=800, max_pivots=40000, reserved=1)
out = prob.solve(z["x0"][i:i + 1], z["omega"][i:i + 1], z["model_idx"][i:i + 1].astype(np.int32))
print("RESULT", out["status"], out["obj"], out["lower_bound"], out["nodes"], out["pivots"])
