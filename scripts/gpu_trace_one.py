"""One instance of a bench scenario set with the solver trace (needs a -DMLD_TRACE build: MLD_CXXFLAGS=-DMLD_TRACE python -m pyhybridcontrol_amd.build --force)
python scripts/gpu_trace_one.py <set> <instance> [reserved bits besides the trace]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from pyhybridcontrol_amd import gpu, host, synthetic as syn
t_set, inst = int(sys.argv[1]), int(sys.argv[2])
extra = int(sys.argv[3]) if len(sys.argv) > 3 else 0
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 512, 0)
d = agents[0]["dims"]
rng = np.random.Generator(np.random.PCG64([syn.CONFIGS["cfg4"]["seed"], 7919, 0, t_set]))
xs, ws = syn.make_scenarios(d["nx"], N_t, x0.shape[0], rng)
a = agents[midx[inst]]
model = gpu.GpuModel([a["mats"]], d)
prob = gpu.GpuProblem(model, N_p, N_t, host.cost_from_atoms(a["atoms"], d, N_p, N_t), gap_rel=1e-2, max_nodes=800, max_pivots=40000, reserved=1 | extra)
out = prob.solve(xs[inst:inst + 1], ws[inst:inst + 1])
sys.stdout.flush()
print("RESULT status %d obj %.9g lb %.9g nodes %d pivots %d" % (out["status"][0], out["obj"][0], out["lower_bound"][0], out["nodes"][0], out["pivots"][0]))
