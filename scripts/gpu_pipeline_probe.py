"""Consecutive, independent scenario sets on TWO problem handles with their own HIP streams (mld_solve_launch / mld_solve_finish) against one
handle solving them one after another: the workgroups of set k+1 move onto the CUs the stragglers of set k leave idle.
python scripts/gpu_pipeline_probe.py [steps] [scenarios] [handles]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from pyhybridcontrol_amd import gpu, host, synthetic as syn
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_scen = int(sys.argv[2]) if len(sys.argv) > 2 else 512
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]["dims"]
model = gpu.GpuModel([a["mats"] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
sets = [(x0, om)]
for t in range(1, K + 2):
    rng = np.random.Generator(np.random.PCG64([syn.CONFIGS["cfg4"]["seed"], 7919, 0, t]))
    sets.append(syn.make_scenarios(d["nx"], N_t, x0.shape[0], rng))
X = np.stack([s[0] for s in sets]); W = np.stack([s[1] for s in sets])
kw = dict(gap_rel=1e-2, max_nodes=800, max_pivots=40000)
probs = [gpu.GpuProblem(model, N_p, N_t, cost, **kw) for _ in range(H)]
for p in probs:
    p.upload(x0, om, midx); p.stage(X, W); p.solve_resident()          # warm-up (also learns a first queue order)
# one after another on one handle
p = probs[0]
t0 = time.perf_counter(); ref = []
for k in range(1, K + 1):
    p.select(k); st = p.solve_resident(); ref.append((st["n_optimal"], st["pivots"]))
serial = time.perf_counter() - t0
obj_serial = p.download()["obj"].copy()
# two handles, two streams
for p in probs: p.use_stream()
t0 = time.perf_counter(); got = [None] * K
for k in range(1, K + 1):
    p = probs[k % H]
    if k > H:
        st = p.finish(); got[k - H - 1] = (st["n_optimal"], st["pivots"])
    p.select(k); p.launch()
for k in range(K - H + 1, K + 1):
    st = probs[k % H].finish(); got[k - 1] = (st["n_optimal"], st["pivots"])
piped = time.perf_counter() - t0
obj_piped = probs[K % H].download()["obj"]
n = x0.shape[0]
print("%d steps of %d instances: one after another %.2f s (%.0f solves/s), %d streams %.2f s (%.0f solves/s): %.2fx" % (K, n, serial, K * n / serial, H, piped, K * n / piped, serial / piped))
print("per-step (proven, pivots) equal: %s; last step's objectives identical: %s" % (ref == got, bool(np.array_equal(obj_serial, obj_piped))))
